"""Capture golden vectors from the reference's OWN modules (build container only).

    python oracle/make_golden.py            # writes tests/golden/*.npz

Imports /root/reference through `oracle/ref_harness.py`, drives the reference modules on
deterministic synthetic inputs (`tests/synth.py`) and stores the OUTPUTS (plus float64
checksums of the inputs) as small fixtures.  The reference never travels: the GPU box
sees only these fixtures.  Fixtures are data, not source.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness  # noqa: E402
from tests import synth         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def save(name, meta, **arrays):
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), **arrays)
    size = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  {name}.npz  {size/1024:.1f} KiB  {list(arrays)}")


# ---------------------------------------------------------------- VQ layer
VQ_CASES = [
    # name, B, C, H, W, K, flavour
    dict(name="vq_small", b=2, c=64, h=8, w=8, k=32, flavour="relu"),
    dict(name="vq_k256", b=2, c=128, h=8, w=8, k=256, flavour="relu"),
    dict(name="vq_real_l2", b=1, c=512, h=16, w=16, k=512, flavour="relu"),
    dict(name="vq_real_l4", b=2, c=2048, h=4, w=4, k=512, flavour="relu"),
    dict(name="vq_ties", b=1, c=64, h=8, w=8, k=64, flavour="ties"),
    dict(name="vq_dead", b=1, c=32, h=4, w=4, k=128, flavour="dead"),
    dict(name="vq_signed", b=2, c=48, h=5, w=7, k=96, flavour="signed"),
]


def vq_inputs(case):
    """Deterministic (x, W, g) for a VQ case.  Shared with the tests."""
    b, c, h, w, k = case["b"], case["c"], case["h"], case["w"], case["k"]
    seed = 1000 + sum(map(ord, case["name"]))
    fl = case["flavour"]
    if fl == "signed":
        x = synth.uniform(seed, (b, c, h, w), -1.0, 1.0)
        W = synth.uniform(seed + 1, (k, c), -1.0, 1.0)
    else:
        x = synth.relu_features(seed, (b, c, h, w))
        W = synth.relu_features(seed + 1, (k, c), sparsity=0.3, scale=1.5)
    if fl == "ties":                       # duplicated codes: the lowest index must win
        W[k // 2:] = W[: k - k // 2]
        rows = x.permute(0, 2, 3, 1).reshape(-1, c)
        rows[:8] = W[5:13]                  # exact hits (distance ~0 after cancellation)
        x = rows.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    if fl == "dead":                       # far-away codes that can never be selected
        W[k // 4:] += 50.0
    g = synth.uniform(seed + 2, (b, c, h, w), -1.0, 1.0)
    return x, W, g


def gen_vq(ref):
    for case in VQ_CASES:
        x, W, g = vq_inputs(case)
        vq = ref.vq_img.VectorQuantizer(dim=case["c"], num_embeddings=case["k"], commitment_weight=1)
        with torch.no_grad():
            vq.codebook.embedding.weight.copy_(W)
        vq.eval()
        with torch.no_grad():
            q_e, idx_e, loss_e, use_e = vq(x)
        vq.train()
        xr = x.clone().requires_grad_(True)
        q_t, idx_t, loss_t, use_t = vq(xr)
        ((q_t * g).sum() + 3.0 * loss_t.sum()).backward()
        # fp64 margin between best and runner-up squared distance (for the near-tie audit)
        rows = x.permute(0, 2, 3, 1).reshape(-1, case["c"]).double()
        d2 = (rows[:, None, :] - W.double()[None]).pow(2).sum(-1)
        top2 = torch.topk(d2, 2, dim=1, largest=False).values
        rel_gap = ((top2[:, 1] - top2[:, 0]) / top2[:, 1].clamp_min(1e-30))
        meta = dict(case, x_sum=synth.checksum(x), w_sum=synth.checksum(W), g_sum=synth.checksum(g),
                    min_rel_gap_fp64=float(rel_gap.min()), grad_loss_scale=3.0,
                    source="vector_quantizer/vq_img.py:193-244 VectorQuantizer")
        save(case["name"], meta, q_eval=q_e, idx_eval=idx_e, loss_eval=loss_e, usage_eval=use_e,
             q_train=q_t, idx_train=idx_t, loss_train=loss_t, usage_train=use_t, grad_x=xr.grad,
             w_grad_is_none=np.array(vq.codebook.embedding.weight.grad is None))


# ---------------------------------------------------------------- argmin at BASELINE row counts, live codebooks
def gen_vq_big(ref):
    """the reference's EuclideanCodebook.forward (vector_quantizer/vq_img.py:160-177) at the row counts of the
    BASELINE configurations, on codebooks where every code is alive.  Per case (tests/cases.py::VQ_BIG_CASES): structured post-ReLU rows;
    the reference's own `kmeans` (10 iterations, initial means patched to fixed sample rows); labels = the fp64 assignment to those
    means; codebook = one more Lloyd update from the labels in float64 with a fixed summation order (cases.codebook_from_labels --
    the test re-derives the SAME bits from the stored labels); then the reference VectorQuantizer in eval mode on (a) the fp32 rows and
    (b) the same rows rounded to bfloat16.  Stored: labels (int16), the indices as sparse differences against the labels, and the
    fp64 top-2 relative-gap histogram of both row sets."""
    from tests import cases
    import time
    arrays, metas = {}, []
    for case in cases.VQ_BIG_CASES:
        t0 = time.time()
        n, c, k = case["n"], case["c"], case["k"]
        rows = cases.vq_big_rows(case)
        means0 = cases.vq_big_means0(rows, case)
        orig = ref.vq_img.batched_sample_vectors
        ref.vq_img.batched_sample_vectors = lambda s_, num, _m=means0: _m[None].clone()
        try:
            means, bins = ref.vq_img.kmeans(rows[None], k, 10)
        finally:
            ref.vq_img.batched_sample_vectors = orig
        assert int((bins[0] == 0).sum()) == 0, "k-means left an empty cluster: pick other initial means"

        def d2_fp64(r, W):
            r64, W64 = r.double(), W.double()
            return (r64.pow(2).sum(1, keepdim=True) + W64.pow(2).sum(1)[None] - 2.0 * r64 @ W64.t()).clamp_min(0)
        labels = d2_fp64(rows, means[0]).argmin(1)
        W = cases.codebook_from_labels(rows, labels.numpy(), k)
        out = {}
        for tag, r in (("f32", rows), ("bf16", cases.bf16_exact(rows))):
            vq = ref.vq_img.VectorQuantizer(dim=c, num_embeddings=k, commitment_weight=1)
            with torch.no_grad():
                vq.codebook.embedding.weight.copy_(W)
            vq.eval()
            x = r.reshape(1, n, 1, c).permute(0, 3, 1, 2)                     # (B=1, C, H=n, W=1): rows in order
            with torch.no_grad():
                q, idx, loss, usage = vq(x)
            idx = idx.reshape(-1)
            assert torch.equal(q.permute(0, 2, 3, 1).reshape(n, c), W[idx])
            d2 = d2_fp64(r, W)
            top2 = torch.topk(d2, 2, dim=1, largest=False)
            gap = (top2.values[:, 1] - top2.values[:, 0]) / top2.values[:, 1].clamp_min(1e-30)
            hist = torch.histc(gap.clamp(1e-12, 1.0).log10(), bins=24, min=-12.0, max=0.0)
            wrong64 = int((top2.indices[:, 0] != idx).sum())
            diff = (idx != labels).nonzero()[:, 0]
            out[tag] = dict(usage=float(usage), min_gap=float(gap.min()), n_gap_lt_1e6=int((gap < 1e-6).sum()), n_gap_lt_1e5=int((gap < 1e-5).sum()),
                            n_differs_from_fp64_argmin=wrong64, n_differs_from_labels=int(diff.numel()))
            arrays[f"{case['name']}/{tag}/diff_pos"] = diff.to(torch.int32)
            arrays[f"{case['name']}/{tag}/diff_idx"] = idx[diff].to(torch.int16)
            arrays[f"{case['name']}/{tag}/gap_hist_log10"] = hist.to(torch.int32)
            arrays[f"{case['name']}/{tag}/idx_sum"] = idx.double().sum()
        arrays[f"{case['name']}/labels"] = labels.to(torch.int16)
        metas.append(dict(case, rows_bits=synth.bits_checksum(rows), w_bits=synth.bits_checksum(W), **{f"{t}_{k_}": v for t, d in out.items() for k_, v in d.items()}))
        print(f"  {case['name']}: {time.time() - t0:.0f}s  {out}")
    save("vq_big", dict(cases=metas, source="vector_quantizer/vq_img.py:29-63 kmeans (patched initial means) -> :193-244 VectorQuantizer eval; "
                                            "codebook = tests/cases.py::codebook_from_labels(labels)"), **arrays)


# ---------------------------------------------------------------- k-means
KMEANS_CASES = [
    dict(name="kmeans_small", n=512, c=32, k=16, iters=10, empty=False),
    dict(name="kmeans_empty", n=256, c=16, k=32, iters=10, empty=True),
    dict(name="kmeans_real", n=2048, c=512, k=256, iters=10, empty=False),
    dict(name="kmeans_proto", n=4096, c=32, k=3, iters=10, empty=False),
]


def kmeans_inputs(case):
    seed = 2000 + sum(map(ord, case["name"]))
    n, c, k = case["n"], case["c"], case["k"]
    samples = synth.relu_features(seed, (n, c))
    pick = torch.floor(synth.uniform(seed + 1, (k,)) * n).long().clamp(max=n - 1)
    means0 = samples[pick].clone()
    if case["empty"]:
        means0[k // 2:] = means0[k // 2:] + 100.0     # unreachable -> empty clusters keep old mean
    return samples, means0


def gen_kmeans(ref):
    for case in KMEANS_CASES:
        samples, means0 = kmeans_inputs(case)
        orig = ref.vq_img.batched_sample_vectors
        ref.vq_img.batched_sample_vectors = lambda s, num, _m=means0: _m[None].clone()   # patch the module object
        try:
            means, bins = ref.vq_img.kmeans(samples[None], case["k"], case["iters"])
        finally:
            ref.vq_img.batched_sample_vectors = orig
        meta = dict(case, samples_sum=synth.checksum(samples), means0_sum=synth.checksum(means0),
                    source="vector_quantizer/vq_img.py:29-63 kmeans (initial means patched)")
        save(case["name"], meta, means=means[0], bins=bins[0])


# ---------------------------------------------------------------- decoder
DEC_CASES = [
    dict(name="decoder_small", enc=(3, 8, 16, 24, 32, 48), dec=(24, 16, 12, 8, 4), b=2, s=64),
    dict(name="decoder_odd", enc=(3, 4, 8, 8, 16, 16), dec=(8, 8, 4, 4, 4), b=1, s=96),
]


def decoder_inputs(case):
    seed = 3000 + sum(map(ord, case["name"]))
    enc, b, s = case["enc"], case["b"], case["s"]
    feats = [synth.relu_features(seed + i, (b, enc[i + 1], s >> (i + 1), s >> (i + 1))) for i in range(5)]
    sd = synth.synth_state_dict(synth.decoder_shapes(enc, case["dec"], prefix=""), seed + 50)
    g = synth.uniform(seed + 99, (b, case["dec"][-1], s // 2, s // 2), -1.0, 1.0)
    return feats, sd, g


def gen_decoder(ref):
    for case in DEC_CASES:
        feats, sd, g = decoder_inputs(case)
        dec = ref.decoder.UnetDecoder(list(case["enc"]), list(case["dec"]))
        dec.load_state_dict(sd)
        dec.eval()
        with torch.no_grad():
            y_eval = dec(*feats)
        dec.train()
        fr = [f.clone().requires_grad_(True) for f in feats]
        y_tr = dec(*fr)
        (y_tr * g).sum().backward()
        post = dec.state_dict()
        meta = dict(name=case["name"], enc=list(case["enc"]), dec=list(case["dec"]), b=case["b"], s=case["s"],
                    source="models/networks/unet/decoder.py:14-39 UnetDecoder")
        arrays = dict(y_eval=y_eval, y_train=y_tr)
        for i, f in enumerate(fr):
            arrays[f"grad_feat{i}"] = f.grad
        arrays["grad_w_first"] = dec.blocks[0][0][0].weight.grad
        arrays["grad_w_last"] = dec.blocks[4][1][0].weight.grad
        arrays["grad_bn_w_last"] = dec.blocks[4][1][1].weight.grad
        arrays["grad_bn_b_last"] = dec.blocks[4][1][1].bias.grad
        arrays["run_mean_first"] = post["blocks.0.0.1.running_mean"]
        arrays["run_var_first"] = post["blocks.0.0.1.running_var"]
        arrays["run_mean_last"] = post["blocks.4.1.1.running_mean"]
        arrays["run_var_last"] = post["blocks.4.1.1.running_var"]
        save(case["name"], meta, **arrays)


# ---------------------------------------------------------------- one decoder block at the benchmark's own size
BLOCK_CASE = dict(name="decoder_block0_b32", cin=2048, cout=1024, b=32, s=16)


def block_inputs(case=BLOCK_CASE):
    """Deepest decoder block at bench scale: 2048 -> 1024 -> 1024 channels on 32 x 16 x 16 pixels (B = 32 at 512 x 512)."""
    from collections import OrderedDict
    cin, cout, b, s = case["cin"], case["cout"], case["b"], case["s"]
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


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.pow(2).sum().item(), t.abs().max().item()])


def gen_decoder_block(ref):
    """double_conv_block (models/networks/unet/decoder.py:7-12) at the channel counts and pixel count the benchmark runs, so that
    the kernels the bench dispatches there (multi-chunk patch-reuse forward / data gradient, nine-tap weight gradient) are
    compared with the reference's own block.  Outputs are stored as strided probes + float64 checksums (33 MB otherwise)."""
    x, sd, g = block_inputs()
    blk = ref.decoder.double_conv_block(BLOCK_CASE["cin"], BLOCK_CASE["cout"])
    blk.load_state_dict(sd)
    blk.eval()
    with torch.no_grad():
        y_eval = blk(x)
    blk.train()
    xr = x.clone().requires_grad_(True)
    y = blk(xr)
    (y * g).sum().backward()
    post = blk.state_dict()
    save(BLOCK_CASE["name"], dict(BLOCK_CASE, x_sum=synth.checksum(x), source="models/networks/unet/decoder.py:7-12 double_conv_block"),
         y_eval=probe(y_eval), y_eval_stats=stats(y_eval), y_train=probe(y), y_train_stats=stats(y),
         grad_x=probe(xr.grad), grad_x_stats=stats(xr.grad),
         grad_w0=probe(blk[0][0].weight.grad), grad_w0_stats=stats(blk[0][0].weight.grad),
         grad_w1=probe(blk[1][0].weight.grad), grad_w1_stats=stats(blk[1][0].weight.grad),
         grad_bn_w1=blk[1][1].weight.grad, grad_bn_b1=blk[1][1].bias.grad, grad_bn_w0=blk[0][1].weight.grad,
         run_mean0=post["0.1.running_mean"], run_var0=post["0.1.running_var"], run_var1=post["1.1.running_var"])


# ---------------------------------------------------------------- prototype losses, dice/CE, metrics, lr
def proto_inputs(seed=4000, b=2, c=32, s=16):
    feat = synth.uniform(seed, (b, c, s, s), -1.0, 1.0)
    gt = synth.labels(seed + 1, (b, 2 * s, 2 * s))
    scores = synth.uniform(seed + 2, (b, 3, 2 * s, 2 * s), -3.0, 3.0)
    protos = synth.uniform(seed + 3, (3, c), -1.0, 1.0)
    entropy = synth.uniform(seed + 4, (b * s * s,), 0.0, 1.1)
    return feat, gt, scores, protos, entropy


def gen_proto(ref):
    feat, gt, scores, protos, entropy = proto_inputs()
    out = {}
    for tag, margin, scale in (("m0", 0.0, 1.0), ("m05", 0.5, 30.0)):
        # v1 (later definition, prototype.py:500-613)
        m1 = ref.prototype.ReliablePrototypeLoss(3, 32, scale=scale, margin=margin, init="normal")
        with torch.no_grad():
            m1.embedding.weight.copy_(protos)
        m1.train()
        fr = feat.clone().requires_grad_(True)
        l1 = m1(fr, gt, percent=80.0, entropy=entropy)
        l1.backward()
        out[f"v1_{tag}_loss"] = l1.detach()
        out[f"v1_{tag}_grad"] = fr.grad
        out[f"v1_{tag}_proto_grad_none"] = np.array(m1.embedding.weight.grad is None)
        # v2 forward only (backward raises on fp32): hard labels and pseudo scores
        for kind, target in (("gt", gt), ("score", scores)):
            m2 = ref.prototype.ReliablePrototypeLossv2(3, 32, scale=scale, margin=margin, init="normal")
            with torch.no_grad():
                m2.embedding.weight.copy_(protos)
            m2.train()
            with torch.no_grad():
                l2 = m2(feat, target, 0.7)
            out[f"v2_{tag}_{kind}_loss"] = l2
            out[f"v2_{tag}_{kind}_proto_after"] = m2.embedding.weight.detach().clone()
    save("prototype", dict(source="models/modules/prototype.py:500-613, :778-888", percent=80.0, th=0.7), **out)


def gen_losses(ref):
    seed = 5000
    pred = synth.uniform(seed, (3, 3, 24, 24), -4.0, 4.0)
    pred2 = synth.uniform(seed + 1, (3, 3, 24, 24), -4.0, 4.0)
    tgt = synth.labels(seed + 2, (3, 24, 24))
    dice = ref.loss.make_loss("dice_loss", 3, ignore_index=255)
    ce = torch.nn.CrossEntropyLoss(ignore_index=255)
    out = {}
    p1 = pred.clone().requires_grad_(True)
    sup = 0.5 * ce(p1, tgt) + dice(p1, tgt)                                  # train:179
    sup.backward()
    out["sup_loss"], out["sup_grad"] = sup.detach(), p1.grad
    # CPS block, train:165-177 (score_mask restated from :43-46 -- the trainer file is not importable)
    pa, pb = pred.clone().requires_grad_(True), pred2.clone().requires_grad_(True)
    ps_a, ps_b = torch.argmax(pa, 1).long(), torch.argmax(pb, 1).long()
    fa = torch.where(torch.softmax(pa, 1).max(1)[0] > 0.7, ps_a, 255)
    fb = torch.where(torch.softmax(pb, 1).max(1)[0] > 0.7, ps_b, 255)
    cps = 0.5 * ce(pa, fb) + 0.5 * ce(pb, fa) + dice(pa, fb) + dice(pb, fa)
    cps.backward()
    out.update(cps_loss=cps.detach(), cps_grad_a=pa.grad, cps_grad_b=pb.grad, filt_a=fa, filt_b=fb)
    meas = ref.measurement.Measurement(3)
    conf = meas._make_confusion_matrix(pred.numpy(), tgt.numpy())
    miou, ious = meas.miou(conf)
    out.update(conf=conf, miou=np.array(miou), ious=np.array(ious))
    sched = ref.lr_schedulers.CosineAnnealingLR(start_lr=1e-4, min_lr=1e-7, total_iters=1000, warmup_steps=0)
    out["lr_table"] = np.array([sched.get_lr(i) for i in range(0, 1001, 50)], dtype=np.float64)
    save("losses_metrics", dict(source="loss/dice_loss.py:5-68; train_vqreptunet1x1v2.py:43-46,165-181; "
                                       "measurement.py:12-62; utils/lr_schedulers.py:103-112", th=0.7), **out)


# ---------------------------------------------------------------- whole models (torchvision-like base)
def model_cfg(ref, name, k=(0, 0, 512, 512, 512), margin=0.0, scale=1.0):
    return ref.AttrDict({"name": name, "params": {
        "encoder_name": "resnet50", "num_classes": 3, "depth": 5,
        "vq_cfg": {"num_embeddings": list(k), "distance": "euclidean", "kmeans_init": True},
        "margin": margin, "scale": scale, "use_feature": False, "encoder_weights": None}})


MODEL_SEED = 77


def model_inputs(b=2, s=64, seed=6000):
    x = synth.uniform(seed, (b, 3, s, s))
    gt = synth.blob_labels(seed + 1, b, s, cell=8)
    scores = synth.uniform(seed + 2, (b, 3, s, s), -3.0, 3.0)
    return x, gt, scores


def set_bn_momentum(model, m):
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = m


def prepare_model(model, x, gt, version):
    """Shared preparation recipe (also executed by the tests on their own side):
    1. encoder BN running stats := batch stats of x (train-mode encoder pass, momentum 1)
    2. codebooks := function of the eval-mode encoder features (install_codebooks)
    3. all BN running stats := batch stats of a full train-mode forward (momentum 1)
    so that activations stay O(1) and eval / train features are close to each other."""
    set_bn_momentum(model, 1.0)
    model.train()
    with torch.no_grad():
        model.encoder(x)
    model.eval()
    install_codebooks(model, x)
    model.train()
    with torch.no_grad():
        if version == 1:
            model(x, gt, percent=80.0)
        elif version == 2:
            model(x, gt, th=0.7)
        else:
            model(x)
    set_bn_momentum(model, 0.1)


def install_codebooks(model, x):
    """Codebooks derived from the model's own eval-mode encoder features (tests/cases.py recipe)."""
    from tests import cases
    with torch.no_grad():
        feats = model.encoder(x)[1:]
        for i in (2, 3, 4):
            cb = model.codebook[i].codebook
            cb.embedding.weight.copy_(cases.codebook_from_rows(cases.rows_of(feats[i]), cb.num_embeddings, 900 + i))
            cb.initted = True


PROBE_KEYS = ["segmentation_head.weight", "encoder.conv1.weight", "decoder.blocks.4.1.0.weight",
              "decoder.blocks.0.0.0.weight", "encoder.layer4.2.conv3.weight", "encoder.layer2.0.downsample.0.weight",
              "encoder.layer1.0.bn1.weight", "decoder.blocks.2.0.1.bias"]


def gen_models(ref, size=64, suffix="", versions=(1, 2)):
    x, gt, scores = model_inputs(s=size)
    for version, name, margin, scale in ((1, "vqreptunet1x1", 0.0, 1.0), (2, "vqreptunet1x1v2", 0.5, 30.0)):
        if version not in versions:
            continue
        torch.manual_seed(0)
        model = ref.networks.make_model(model_cfg(ref, name, margin=margin, scale=scale))
        shapes = synth.shapes_of(model.state_dict())
        sd = synth.synth_state_dict(shapes, MODEL_SEED)
        model.load_state_dict(sd)
        model.prototype_loss.initted = True
        prepare_model(model, x, gt, version)
        model.eval()
        out = {}
        with torch.no_grad():
            o = model(x)
        out.update(eval_logits=o[0], eval_loss=o[1], eval_usage=o[2])
        # indices per level in eval mode
        with torch.no_grad():
            feats = model.encoder(x)[1:]
            for i in (2, 3, 4):
                out[f"eval_idx{i}"] = model.codebook[i](feats[i])[1]
            out["eval_feat_sums"] = np.array([synth.checksum(f) for f in feats])
        model.train()
        if version == 1:
            logits, closs, usage, ploss = model(x, gt, percent=80.0)
            total = (logits * synth.uniform(6100, tuple(logits.shape), -1.0, 1.0)).sum() + 2.0 * closs.sum() + 5.0 * ploss
        else:
            logits, closs, usage, ploss = model(x, gt, th=0.7)
            total = (logits * synth.uniform(6100, tuple(logits.shape), -1.0, 1.0)).sum() + 2.0 * closs.sum()
            with torch.no_grad():
                model2 = ref.networks.make_model(model_cfg(ref, name, margin=margin, scale=scale))
                model2.load_state_dict(model.state_dict())
                for i in (2, 3, 4):
                    model2.codebook[i].codebook.initted = True
                model2.prototype_loss.initted = True
                model2.train()
                out["train_proto_score"] = model2(x, scores, th=0.7)[3]
        total.backward()
        out.update(train_logits=logits, train_loss=closs, train_usage=usage, train_proto=ploss)
        named = dict(model.named_parameters())
        for key in PROBE_KEYS:
            out["grad/" + key] = probe(named[key].grad)
            out["gradnorm/" + key] = named[key].grad.double().norm()
        out["grad_none_keys"] = np.array([k for k, p in named.items() if p.grad is None])
        post = model.state_dict()
        for key in ("encoder.bn1.running_mean", "encoder.bn1.running_var", "encoder.layer4.2.bn3.running_var",
                    "decoder.blocks.4.1.1.running_mean", "decoder.blocks.0.0.1.running_var"):
            out["post/" + key] = post[key]
        meta = dict(version=version, name=name, margin=margin, scale=scale, model_seed=MODEL_SEED, size=size,
                    n_keys=len(shapes), n_params=int(sum(p.numel() for p in model.parameters())),
                    key_shapes_digest=hash_shapes(shapes), percent=80.0, th=0.7, loss_scale=2.0, proto_scale=5.0,
                    source="models/networks/modified_vqunet/net.py:1141-1222 / :184-260 on a torchvision-like ResNet base")
        save(f"model_v{version}{suffix}", meta, **out)
        if version == 1 and not suffix:
            with open(os.path.join(OUT, "state_dict_layout_vqreptunet1x1.json"), "w") as f:
                json.dump({k: list(v) for k, v in shapes.items()}, f)


def gen_models128(ref):
    """The v1 model at 128^2: 32 samples per channel at the deepest level (64^2 has 8), which is what lets the GPU test hold
    the whole-model gradients to a tighter bar (train-mode BatchNorm over a handful of samples amplifies rounding)."""
    gen_models(ref, size=128, suffix="_128", versions=(1,))


def gen_unet(ref):
    cfg = ref.AttrDict({"name": "unet", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                                    "encoder_weights": None}})
    model = ref.networks.make_model(cfg)
    shapes = synth.shapes_of(model.state_dict())
    sd = synth.synth_state_dict(shapes, MODEL_SEED + 1)
    model.load_state_dict(sd)
    x, gt, _ = model_inputs(b=2, s=64, seed=6500)
    set_bn_momentum(model, 1.0)
    model.train()
    with torch.no_grad():
        model(x)
    set_bn_momentum(model, 0.1)
    model.eval()
    with torch.no_grad():
        y_eval = model(x)
    model.train()
    y = model(x)
    dice = ref.loss.make_loss("dice_loss", 3, ignore_index=255)
    loss = dice(y, gt) + 0.5 * F.cross_entropy(y, gt, ignore_index=255)     # deprecated/train_baseline.py:128-140
    loss.backward()
    named = dict(model.named_parameters())
    out = dict(eval_logits=y_eval, train_logits=y, loss=loss.detach())
    for key in ("segmentation_head.0.weight", "segmentation_head.0.bias", "encoder.conv1.weight", "decoder.blocks.4.1.0.weight"):
        out["grad/" + key] = probe(named[key].grad)
        out["gradnorm/" + key] = named[key].grad.double().norm()
    save("model_unet", dict(model_seed=MODEL_SEED + 1, n_keys=len(shapes), key_shapes_digest=hash_shapes(shapes),
                            source="models/networks/unet/net.py:806-838 Unet"), **out)
    with open(os.path.join(OUT, "state_dict_layout_unet.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f)


# ---------------------------------------------------------------- CPS iterations of the trainers (SURVEY 8c fixture (9))
def _ref_ns(ref):
    import types
    return types.SimpleNamespace(models=ref.models, make_loss=ref.loss.make_loss, Measurement=ref.measurement.Measurement,
                                 CosineAnnealingLR=ref.lr_schedulers.CosineAnnealingLR)


def gen_cps(ref):
    """Two iterations of the trainers' loop bodies on the reference's own modules (tests/cps_loop.py restates the loop
    body once; the trainer files are not importable): v1 = deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203 with
    backward + Adam; v2 = train_vqreptunet1x1v2.py:137-196 forward terms only (its backward raises on fp32)."""
    from tests import cps_loop
    dev = torch.device("cpu")
    for version, backward in ((1, True), (2, False)):
        outs = cps_loop.run_iterations(_ref_ns(ref), version, dev, n_iters=2, backward=backward, to_cfg=ref.AttrDict,
                                       prepare=prepare_model)
        arrays = {}
        for i, o in enumerate(outs):
            for k, v in o.items():
                if k.startswith("grad_none"):
                    arrays[f"it{i}/{k}"] = np.array(v)
                elif k.startswith("param/"):
                    arrays[k] = v
                elif isinstance(v, float):
                    arrays[f"it{i}/{k}"] = np.array(v, dtype=np.float64)
                else:
                    arrays[f"it{i}/{k}"] = v
        meta = dict(version=version, size=cps_loop.SIZE, batch=cps_loop.BATCH, seeds=list(cps_loop.SEEDS), train=cps_loop.TRAIN,
                    backward=backward, n_iters=2,
                    source=("deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203" if version == 1 else
                            "train_vqreptunet1x1v2.py:137-196 (forward terms; backward raises in the reference, q10)") +
                           " restated in tests/cps_loop.py, driven on the reference's models / loss / measurement / utils modules")
        save(f"cps_iter_v{version}", meta, **arrays)


def gen_curve(ref):
    """The CPU side of the mIoU-parity run: tests/cps_loop.py::run_curve on the reference's modules (v1 recipe)."""
    from tests import cps_loop
    import time
    t0 = time.time()
    out = cps_loop.run_curve(_ref_ns(ref), torch.device("cpu"), to_cfg=ref.AttrDict, prepare=prepare_model)
    print(f"  curve: {time.time() - t0:.0f}s  test mIoU {out['test_miou']}")
    save("cps_curve_v1", dict(spec=cps_loop.CURVE, k=[0, 0, 64, 64, 64],
                              source="tests/cps_loop.py::run_curve on the reference's modules (v1 recipe, fp32 CPU)"), **out)


def gen_curve128(ref):
    """r4: the thicker mIoU-parity run: 128x128, K = 512 at the three levels, 200 v1 iterations on the
    reference's modules, test mIoU every 25 steps."""
    from tests import cps_loop
    import time
    t0 = time.time()
    out = cps_loop.run_curve(_ref_ns(ref), torch.device("cpu"), to_cfg=ref.AttrDict, prepare=prepare_model, k=cps_loop.K128,
                             spec=cps_loop.CURVE128)
    print(f"  curve128: {time.time() - t0:.0f}s  test mIoU {out['test_miou']}")
    save("cps_curve_v1_128", dict(spec=cps_loop.CURVE128, k=list(cps_loop.K128),
                                  source="tests/cps_loop.py::run_curve on the reference's modules (v1 recipe, fp32 CPU)"), **out)


def probe(t, limit=16384, take=4096):
    """Large tensors are stored as a strided sample (fixtures stay small)."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= limit:
        return t.detach()
    stride = flat.numel() // take
    return flat[::stride][:take].clone()


def hash_shapes(shapes):
    import hashlib
    return hashlib.sha256(json.dumps([[k, list(v)] for k, v in shapes.items()]).encode()).hexdigest()[:16]


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_harness.ref_modules()
    which = set(sys.argv[1:]) or {"vq", "kmeans", "decoder", "proto", "losses", "models", "models128", "unet", "cps", "curve", "block"}
    for tag, fn in (("vq", gen_vq), ("kmeans", gen_kmeans), ("decoder", gen_decoder), ("proto", gen_proto),
                    ("losses", gen_losses), ("models", gen_models), ("models128", gen_models128), ("unet", gen_unet), ("cps", gen_cps), ("curve", gen_curve), ("curve128", gen_curve128), ("block", gen_decoder_block), ("vqbig", gen_vq_big)):
        if tag in which:
            print(f"[{tag}]")
            fn(ref)


if __name__ == "__main__":
    main()
