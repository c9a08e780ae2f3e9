"""Deterministic inputs for every golden case (the same recipes oracle/make_golden.py used).

Kept free of any reference import so it runs on the GPU box.
"""
from __future__ import annotations

import torch

from tests import synth

VQ_CASES = ["vq_small", "vq_k256", "vq_real_l2", "vq_real_l4", "vq_ties", "vq_dead", "vq_signed"]
KMEANS_CASES = ["kmeans_small", "kmeans_empty", "kmeans_real", "kmeans_proto"]
DEC_CASES = ["decoder_small", "decoder_odd"]
MODEL_SEED = 77


def vq_inputs(meta):
    b, c, h, w, k = meta["b"], meta["c"], meta["h"], meta["w"], meta["k"]
    seed = 1000 + sum(map(ord, meta["name"]))
    fl = meta["flavour"]
    if fl == "signed":
        x = synth.uniform(seed, (b, c, h, w), -1.0, 1.0)
        W = synth.uniform(seed + 1, (k, c), -1.0, 1.0)
    else:
        x = synth.relu_features(seed, (b, c, h, w))
        W = synth.relu_features(seed + 1, (k, c), sparsity=0.3, scale=1.5)
    if fl == "ties":
        W[k // 2:] = W[: k - k // 2]
        rows = x.permute(0, 2, 3, 1).reshape(-1, c)
        rows[:8] = W[5:13]
        x = rows.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    if fl == "dead":
        W[k // 4:] += 50.0
    g = synth.uniform(seed + 2, (b, c, h, w), -1.0, 1.0)
    assert synth.checksum(x) == meta["x_sum"] and synth.checksum(W) == meta["w_sum"], "synthetic inputs drifted"
    return x, W, g


def kmeans_inputs(meta):
    seed = 2000 + sum(map(ord, meta["name"]))
    n, c, k = meta["n"], meta["c"], meta["k"]
    samples = synth.relu_features(seed, (n, c))
    pick = torch.floor(synth.uniform(seed + 1, (k,)) * n).long().clamp(max=n - 1)
    means0 = samples[pick].clone()
    if meta["empty"]:
        means0[k // 2:] = means0[k // 2:] + 100.0
    assert synth.checksum(samples) == meta["samples_sum"] and synth.checksum(means0) == meta["means0_sum"]
    return samples, means0


def decoder_inputs(meta):
    seed = 3000 + sum(map(ord, meta["name"]))
    enc, b, s = meta["enc"], meta["b"], meta["s"]
    feats = [synth.relu_features(seed + i, (b, enc[i + 1], s >> (i + 1), s >> (i + 1))) for i in range(5)]
    sd = synth.synth_state_dict(synth.decoder_shapes(enc, meta["dec"], prefix=""), seed + 50)
    g = synth.uniform(seed + 99, (b, meta["dec"][-1], s // 2, s // 2), -1.0, 1.0)
    return feats, sd, g


def proto_inputs(seed=4000, b=2, c=32, s=16):
    feat = synth.uniform(seed, (b, c, s, s), -1.0, 1.0)
    gt = synth.labels(seed + 1, (b, 2 * s, 2 * s))
    scores = synth.uniform(seed + 2, (b, 3, 2 * s, 2 * s), -3.0, 3.0)
    protos = synth.uniform(seed + 3, (3, c), -1.0, 1.0)
    entropy = synth.uniform(seed + 4, (b * s * s,), 0.0, 1.1)
    return feat, gt, scores, protos, entropy


def loss_inputs(seed=5000):
    pred = synth.uniform(seed, (3, 3, 24, 24), -4.0, 4.0)
    pred2 = synth.uniform(seed + 1, (3, 3, 24, 24), -4.0, 4.0)
    tgt = synth.labels(seed + 2, (3, 24, 24))
    return pred, pred2, tgt


def model_inputs(b=2, s=64, seed=6000):
    x = synth.uniform(seed, (b, 3, s, s))
    gt = synth.blob_labels(seed + 1, b, s, cell=8)
    scores = synth.uniform(seed + 2, (b, 3, s, s), -3.0, 3.0)
    return x, gt, scores


def logits_cotangent(shape):
    return synth.uniform(6100, tuple(shape), -1.0, 1.0)
