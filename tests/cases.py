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


# ---------------------------------------------------------------- argmin at BASELINE row counts on LIVE codebooks (vq_big.npz)
VQ_BIG_CASES = [
    # B = 8 images of 512x512: the three levels of one forward (also run as ONE grouped launch), then K = 256 / K = 1024
    dict(name="l2_k512", n=32768, c=512, k=512), dict(name="l3_k512", n=8192, c=1024, k=512), dict(name="l4_k512", n=2048, c=2048, k=512),
    dict(name="l2_k256", n=8192, c=512, k=256), dict(name="l2_k1024", n=16384, c=512, k=1024),
]


def vq_big_rows(case) -> torch.Tensor:
    """Post-ReLU pixel rows with cluster structure: iid post-ReLU noise on top of k sparse post-ReLU centres at a quarter of its
    scale (row i sits near centre i mod k: overlapping clusters, so that second-best codes come close).  Only IEEE-exact fp32
    operations -> identical on every machine."""
    n, c, k = case["n"], case["c"], case["k"]
    seed = 7000 + sum(map(ord, case["name"]))
    base = synth.relu_features(seed, (n, c))
    cent = synth.relu_features(seed + 3, (k, c), sparsity=0.5, scale=1.0)
    rows = base + cent[torch.arange(n) % k] * case.get("centre_scale", 0.25)
    # values on a 2^-12 grid (exact fp32 operations): every cluster sum is then EXACT in float64 whatever the summation order, so
    # codebook_from_labels() returns the same bits on every machine / numpy build
    return (torch.round(rows * 4096.0) / 4096.0).contiguous()


def vq_big_means0(rows: torch.Tensor, case) -> torch.Tensor:
    """Initial means for the reference's k-means: the first row of every centre (no cluster can start empty)."""
    return rows[: case["k"]].clone()


def codebook_from_labels(rows: torch.Tensor, labels, k: int) -> torch.Tensor:
    """One Lloyd update in float64, rounded to fp32: the codebook both the fixture generator (reference side) and the GPU test derive
    from the stored cluster labels.  The rows sit on a 2^-12 grid (vq_big_rows), so the float64 sums are exact and the result is
    bit-identical on every machine -- the reference's indices for it can be compared exactly.  Every cluster must be non-empty."""
    import numpy as np
    lab = np.asarray(labels).astype(np.int64)
    sums = np.zeros((k, rows.shape[1]), dtype=np.float64)
    np.add.at(sums, lab, rows.numpy().astype(np.float64))
    cnt = np.bincount(lab, minlength=k)
    assert (cnt > 0).all(), "dead cluster in a live-codebook fixture"
    return torch.from_numpy((sums / cnt[:, None]).astype(np.float32))


def bf16_exact(rows: torch.Tensor) -> torch.Tensor:
    """fp32 rows whose every value is a bfloat16 (round-to-nearest-even): what the bf16 training path hands to the VQ layer."""
    return rows.bfloat16().float()


def vq_big_expected(fx, case, tag: str):
    """(rows fp32, codebook, expected reference indices int64) of one vq_big case; tag "f32" | "bf16" (rows rounded to bfloat16)."""
    name = case["name"]
    rows = vq_big_rows(case)
    assert synth.bits_checksum(rows) == case["rows_bits"], "synthetic rows drifted"
    labels = fx[f"{name}/labels"].long()
    W = codebook_from_labels(rows, labels.numpy(), case["k"])
    assert synth.bits_checksum(W) == case["w_bits"], "codebook_from_labels is not bit-reproducible on this machine"
    idx = labels.clone()
    idx[fx[f"{name}/{tag}/diff_pos"].long()] = fx[f"{name}/{tag}/diff_idx"].long()
    assert float(idx.double().sum()) == float(fx[f"{name}/{tag}/idx_sum"])
    return (bf16_exact(rows) if tag == "bf16" else rows), W, idx


def near_tie_audit(rows: torch.Tensor, W: torch.Tensor, got: torch.Tensor, want: torch.Tensor):
    """Rows whose index differs from the expected one, audited in float64: (count, largest relative squared-distance gap between the
    two candidate codes over those rows, largest excess of either candidate over the true fp64 minimum).  SURVEY 7: an fp32 distance
    carries ~1e-6 of rounding whose sign depends on the accumulation order (ATen's blocked sgemm on the CPU, a k-ordered fmaf chain
    on the MFMA path), so two correct fp32 implementations may order a pair of codes closer than that differently -- those rows, and
    only those, may differ."""
    bad = (got != want).nonzero()[:, 0]
    if bad.numel() == 0:
        return 0, 0.0, 0.0
    d2 = (rows[bad].double()[:, None, :] - W.double()[None]).pow(2).sum(-1)
    dg, dw = d2.gather(1, got[bad, None])[:, 0], d2.gather(1, want[bad, None])[:, 0]
    best = d2.min(1).values
    gap = (dg - dw).abs() / torch.maximum(dg, dw).clamp_min(1e-30)
    excess = (torch.maximum(dg, dw) - best) / best.clamp_min(1e-30)
    return int(bad.numel()), float(gap.max()), float(excess.max())
