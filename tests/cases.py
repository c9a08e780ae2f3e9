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


def codebook_from_rows(rows: torch.Tensor, k: int, seed: int) -> torch.Tensor:
    """Codebook with WELL SEPARATED winners for model-level fixtures: the first N codes are copies of
    the N feature rows (in a scrambled order), later codes are increasingly perturbed copies.  Built from
    whatever features the caller's own encoder produced, so tiny cross-device rounding differences in
    the features move the codes along with them instead of flipping near-tied argmins."""
    n, c = rows.shape
    ar = torch.arange(k)
    pick = (ar * 7) % n
    mult = (ar // n).float()[:, None]
    u = synth.uniform(seed, (k, c), -1.0, 1.0)
    rms = rows.detach().float().pow(2).mean().sqrt()            # perturbation relative to the feature scale
    return (rows.detach()[pick.to(rows.device)] + (0.5 * mult * u).to(rows.device) * rms).contiguous()


def rows_of(feat: torch.Tensor) -> torch.Tensor:
    return feat.permute(0, 2, 3, 1).reshape(-1, feat.shape[1])


def set_bn_momentum(model, m):
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = m


def prepare_module_model(model, x, gt, version, to_input=lambda t: t):
    """The preparation recipe of oracle/make_golden.py::prepare_model for an nn.Module model:
    calibrate encoder BN (momentum 1) -> codebooks from eval features -> calibrate all BN."""
    set_bn_momentum(model, 1.0)
    model.train()
    with torch.no_grad():
        model.encoder(to_input(x))
    model.eval()
    with torch.no_grad():
        feats = model.encoder(to_input(x))[1:]
        for i in (2, 3, 4):
            cb = model.codebook[i].codebook
            cb.embedding.weight.copy_(codebook_from_rows(rows_of(feats[i]), cb.num_embeddings, 900 + i))
            cb.initted = True
    model.train()
    with torch.no_grad():
        if version == 1:
            model(x, gt, percent=80.0)
        else:
            model(x, gt, th=0.7)
    set_bn_momentum(model, 0.1)


def block_inputs(meta):
    """oracle/make_golden.py::block_inputs: one decoder block at bench scale (2048 -> 1024 -> 1024 channels, 32 x 16 x 16 pixels)."""
    from collections import OrderedDict
    cin, cout, b, s = meta["cin"], meta["cout"], meta["b"], meta["s"]
    shapes = OrderedDict()
    for j, ci in enumerate((cin, cout)):
        shapes[f"{j}.0.weight"] = (cout, ci, 3, 3)
        shapes[f"{j}.1.weight"] = (cout,)
        shapes[f"{j}.1.bias"] = (cout,)
        shapes[f"{j}.1.running_mean"] = (cout,)
        shapes[f"{j}.1.running_var"] = (cout,)
        shapes[f"{j}.1.num_batches_tracked"] = ()
    x = synth.relu_features(3500, (b, cin, s, s))
    sd = synth.synth_state_dict(shapes, 3501)
    g = synth.uniform(3502, (b, cout, s, s), -1.0, 1.0)
    return x, sd, g
