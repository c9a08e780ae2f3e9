"""Pin the CPU oracle (oracle/torch_ref.py) against golden vectors captured from the reference.

CPU only.  Index / integer outputs must match exactly; fp32 outputs within the tolerance
written at each assert (the reference fixtures were produced on another host's BLAS, so
fp32 sums may differ in the last bits).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as R
from tests import cases, golden_io, synth

RTOL = 1e-5     # fp32 restatement vs reference on (possibly) another CPU
ATOL = 1e-6


def close(a, b, rtol=RTOL, atol=ATOL):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"max abs err {err:.3e} (ref max {b.abs().max().item():.3e})"


@pytest.mark.parametrize("name", cases.VQ_CASES)
def test_vq_forward_backward(name):
    fx = golden_io.load(name)
    x, W, g = cases.vq_inputs(fx.meta)
    q, idx, loss, usage = R.vq_forward(x, W, training=False)
    assert torch.equal(idx, fx["idx_eval"])                  # bit-exact indices
    assert torch.equal(q, fx["q_eval"])                      # exact codebook rows (one-hot matmul)
    assert loss.item() == 0.0 and float(usage) == float(fx["usage_eval"])
    xr = x.clone().requires_grad_(True)
    q, idx, loss, usage = R.vq_forward(xr, W, training=True)
    assert torch.equal(idx, fx["idx_train"])
    close(q, fx["q_train"], rtol=1e-6, atol=1e-6)            # x + (q - x): one rounding each way
    close(loss, fx["loss_train"], rtol=1e-5)
    ((q * g).sum() + fx.meta["grad_loss_scale"] * loss.sum()).backward()
    close(xr.grad, fx["grad_x"], rtol=1e-5, atol=1e-7)
    # the analytic backward used by the HIP path agrees with autograd
    ana = R.vq_backward(x, q.detach(), g, torch.tensor(fx.meta["grad_loss_scale"]), 1.0)
    close(ana, fx["grad_x"], rtol=1e-5, atol=1e-7)
    assert bool(fx["w_grad_is_none"])                        # codebook receives no gradient


@pytest.mark.parametrize("name", cases.KMEANS_CASES)
def test_kmeans(name):
    fx = golden_io.load(name)
    samples, means0 = cases.kmeans_inputs(fx.meta)
    means, bins = R.kmeans_lloyd(samples, means0, fx.meta["iters"])
    assert torch.equal(bins, fx["bins"])
    close(means, fx["means"], rtol=1e-5, atol=1e-6)
    if fx.meta["empty"]:
        assert (bins == 0).any() and torch.equal(means[bins == 0], means0[bins == 0])   # vq_img.py:58-61


@pytest.mark.parametrize("name", cases.DEC_CASES)
def test_decoder(name):
    fx = golden_io.load(name)
    feats, sd, g = cases.decoder_inputs(fx.meta)
    p = {"decoder." + k: v.clone() for k, v in sd.items()}
    y = R.unet_decoder(p, feats, training=False)
    close(y, fx["y_eval"], rtol=1e-4, atol=1e-5)
    for k in p:
        if p[k].is_floating_point() and "running" not in k:
            p[k].requires_grad_(True)
    fr = [f.clone().requires_grad_(True) for f in feats]
    y = R.unet_decoder(p, fr, training=True)
    close(y, fx["y_train"], rtol=1e-4, atol=1e-5)
    (y * g).sum().backward()
    for i, f in enumerate(fr):
        close(f.grad, fx[f"grad_feat{i}"], rtol=1e-3, atol=1e-4)
    close(p["decoder.blocks.0.0.0.weight"].grad, fx["grad_w_first"], rtol=1e-3, atol=1e-4)
    close(p["decoder.blocks.4.1.0.weight"].grad, fx["grad_w_last"], rtol=1e-3, atol=1e-4)
    close(p["decoder.blocks.4.1.1.weight"].grad, fx["grad_bn_w_last"], rtol=1e-3, atol=1e-4)
    close(p["decoder.blocks.4.1.1.bias"].grad, fx["grad_bn_b_last"], rtol=1e-3, atol=1e-4)
    close(p["decoder.blocks.0.0.1.running_mean"], fx["run_mean_first"], rtol=1e-5)
    close(p["decoder.blocks.0.0.1.running_var"], fx["run_var_first"], rtol=1e-5)
    close(p["decoder.blocks.4.1.1.running_mean"], fx["run_mean_last"], rtol=1e-5)
    close(p["decoder.blocks.4.1.1.running_var"], fx["run_var_last"], rtol=1e-5)


def test_prototype_losses():
    fx = golden_io.load("prototype")
    feat, gt, scores, protos, entropy = cases.proto_inputs()
    for tag, margin, scale in (("m0", 0.0, 1.0), ("m05", 0.5, 30.0)):
        fr = feat.clone().requires_grad_(True)
        l1 = R.prototype_loss_v1(fr, gt, protos, fx.meta["percent"], entropy, margin, scale)
        assert l1.dtype == torch.float64                                  # q11
        close(l1, fx[f"v1_{tag}_loss"], rtol=1e-6)
        l1.backward()
        close(fr.grad, fx[f"v1_{tag}_grad"], rtol=1e-4, atol=1e-7)
        assert bool(fx[f"v1_{tag}_proto_grad_none"])
        for kind, target in (("gt", gt), ("score", scores)):
            l2, proto_n = R.prototype_loss_v2(feat, target, protos, fx.meta["th"], margin, scale)
            close(l2, fx[f"v2_{tag}_{kind}_loss"], rtol=1e-5)
            close(proto_n, fx[f"v2_{tag}_{kind}_proto_after"], rtol=1e-6)
    # the out-of-place v2 restatement is differentiable (the reference's in-place form is not, q10)
    fr = feat.clone().requires_grad_(True)
    pr = protos.clone().requires_grad_(True)
    R.prototype_loss_v2(fr, gt, pr, 0.7, 0.5, 30.0)[0].backward()
    assert torch.isfinite(fr.grad).all() and torch.isfinite(pr.grad).all()


def test_losses_metrics_schedule():
    fx = golden_io.load("losses_metrics")
    pred, pred2, tgt = cases.loss_inputs()
    p1 = pred.clone().requires_grad_(True)
    sup = 0.5 * F.cross_entropy(p1, tgt, ignore_index=255) + R.dice_loss(p1, tgt)
    close(sup, fx["sup_loss"], rtol=1e-6)
    sup.backward()
    close(p1.grad, fx["sup_grad"], rtol=1e-4, atol=1e-8)
    pa, pb = pred.clone().requires_grad_(True), pred2.clone().requires_grad_(True)
    fa = R.score_mask(pa, torch.argmax(pa, 1).long(), fx.meta["th"])
    fb = R.score_mask(pb, torch.argmax(pb, 1).long(), fx.meta["th"])
    assert torch.equal(fa, fx["filt_a"]) and torch.equal(fb, fx["filt_b"])
    ce = lambda a, b: F.cross_entropy(a, b, ignore_index=255)
    cps = 0.5 * ce(pa, fb) + 0.5 * ce(pb, fa) + R.dice_loss(pa, fb) + R.dice_loss(pb, fa)
    close(cps, fx["cps_loss"], rtol=1e-6)
    cps.backward()
    close(pa.grad, fx["cps_grad_a"], rtol=1e-4, atol=1e-8)
    close(pb.grad, fx["cps_grad_b"], rtol=1e-4, atol=1e-8)
    conf = R.confusion_matrix(pred.numpy(), tgt.numpy())
    assert np.array_equal(conf, fx["conf"].numpy())
    m, ious = R.miou(conf)
    assert m == pytest.approx(float(fx["miou"]), rel=1e-12)
    assert np.allclose(ious, fx["ious"].numpy(), rtol=1e-12)
    lrs = [R.cosine_lr(i, 1e-4, 1e-7, 1000, 0) for i in range(0, 1001, 50)]
    assert np.allclose(lrs, fx["lr_table"].numpy(), rtol=1e-15)


def _model_params(name, seed):
    shapes = golden_io.layout(name)
    return synth.synth_state_dict(shapes, seed)


@pytest.mark.parametrize("version", [1, 2])
def test_whole_model(version):
    """Everything the reference wrote (glue, VQ, decoder, losses) on top of the unpinned ResNet body."""
    fx = golden_io.load(f"model_v{version}")
    sd = _model_params("vqreptunet1x1", fx.meta["model_seed"])       # v1 and v2 share the layout
    assert len(sd) == fx.meta["n_keys"] == 383
    x, gt, scores = cases.model_inputs()
    ks = (0, 0, 512, 512, 512)
    kw0 = dict(percent=80.0) if version == 1 else dict(th=0.7)
    with torch.no_grad():                                           # the preparation recipe of make_golden.prepare_model
        R.resnet_encoder(sd, x, True, momentum=1.0)
        feats = R.resnet_encoder(sd, x, False)[1:]
        for i in (2, 3, 4):
            sd[f"codebook.{i}.codebook.embedding.weight"] = cases.codebook_from_rows(cases.rows_of(feats[i]), 512, 900 + i)
        R.vq_unet_forward(sd, x, True, ks, gt=gt, version=version, margin=fx.meta["margin"], scale=fx.meta["scale"],
                          momentum=1.0, **kw0)
    p = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        logits, closs, usage, proto, aux = R.vq_unet_forward(p, x, False, ks, version=version)
    for j, lvl in enumerate((2, 3, 4)):
        assert torch.equal(aux["indices"][j], fx[f"eval_idx{lvl}"])
    close(logits, fx["eval_logits"], rtol=1e-3, atol=1e-4)             # north_star: 1e-3 relative
    close(usage, fx["eval_usage"], rtol=1e-6)
    assert proto is None and closs.item() == 0.0
    # training forward + backward
    p = {k: v.clone() for k, v in sd.items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k and "codebook" not in k:
            v.requires_grad_(True)
    kw = dict(percent=fx.meta["percent"]) if version == 1 else dict(th=fx.meta["th"])
    logits, closs, usage, proto, aux = R.vq_unet_forward(p, x, True, ks, gt=gt, version=version,
                                                         margin=fx.meta["margin"], scale=fx.meta["scale"], **kw)
    close(logits, fx["train_logits"], rtol=1e-3, atol=1e-4)
    close(closs, fx["train_loss"], rtol=1e-4)
    close(usage, fx["train_usage"], rtol=1e-6)
    close(proto, fx["train_proto"], rtol=1e-4)
    total = (logits * cases.logits_cotangent(logits.shape)).sum() + fx.meta["loss_scale"] * closs.sum()
    if version == 1:
        total = total + fx.meta["proto_scale"] * proto
    total.backward()
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        got = golden_io.probe(p[key].grad)
        ref = fx["grad/" + key]
        scale = ref.abs().max().item() + 1e-12
        assert (got - ref).abs().max().item() <= 2e-3 * scale, key
        assert p[key].grad.double().norm().item() == pytest.approx(float(fx["gradnorm/" + key]), rel=2e-3)
    for key in [k[5:] for k in fx if k.startswith("post/")]:
        close(p[key], fx["post/" + key], rtol=1e-4, atol=1e-6)
    if version == 2:
        with torch.no_grad():
            p2 = {k: v.clone() for k, v in sd.items()}
            out = R.vq_unet_forward(p2, x, True, ks, gt=scores, version=2, th=fx.meta["th"],
                                    margin=fx.meta["margin"], scale=fx.meta["scale"])
        close(out[3], fx["train_proto_score"], rtol=1e-4)
    none_keys = set(fx["grad_none_keys"].tolist())
    assert {f"codebook.{i}.codebook.embedding.weight" for i in (2, 3, 4)} <= none_keys


def test_plain_unet():
    fx = golden_io.load("model_unet")
    sd = synth.synth_state_dict(golden_io.layout("unet"), fx.meta["model_seed"])
    x, gt, _ = cases.model_inputs(b=2, s=64, seed=6500)
    with torch.no_grad():
        R.unet_forward(sd, x, True, momentum=1.0)                  # BN calibration, as in make_golden.gen_unet
    p = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        y = R.unet_forward(p, x, False)
    close(y, fx["eval_logits"], rtol=1e-3, atol=1e-4)
    p = {k: v.clone() for k, v in sd.items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    y = R.unet_forward(p, x, True)
    close(y, fx["train_logits"], rtol=1e-3, atol=1e-4)
    loss = R.dice_loss(y, gt) + 0.5 * F.cross_entropy(y, gt, ignore_index=255)
    close(loss, fx["loss"], rtol=1e-5)
    loss.backward()
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        ref = fx["grad/" + key]
        got = golden_io.probe(p[key].grad)
        assert (got - ref).abs().max().item() <= 2e-3 * (ref.abs().max().item() + 1e-12), key


def test_evaluation_loop_matches_measurement_on_numpy():
    """vq_seg_amd.evaluate.test_loop (device-side confusion counts) against Measurement.measure on the resized logits,
    batch by batch, as the reference's test_detailviz.py:107-123 does."""
    import numpy as np
    import torch.nn.functional as F
    from vq_seg_amd.evaluate import test_loop
    from vq_seg_amd.measurement import Measurement

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.c = torch.nn.Conv2d(3, 3, 3, padding=1)

        def forward(self, x):
            return self.c(x), None, None, None

    model = Tiny()
    batches = [(synth.uniform(i, (2, 3, 24, 20), 0, 1), (synth.uniform(50 + i, (2, 31, 27), 0, 1) * 3).long().clamp_(0, 2)) for i in range(3)]
    got = test_loop(model, batches, 3)
    meas = Measurement(3)
    acc = miou = prec = rec = f1 = 0.0
    ious = np.zeros(3)
    with torch.no_grad():
        for img, tgt in batches:
            pred = F.interpolate(model(img)[0], tgt.shape[-2:], mode="bilinear").numpy()
            a, m, i, p, r, f = meas(pred, tgt.numpy())
            acc, miou, prec, rec, f1, ious = acc + a, miou + m, prec + p, rec + r, f1 + f, ious + np.array(i)
    n = len(batches)
    assert abs(got["test_acc"] - acc / n) < 1e-12 and abs(got["test_miou"] - miou / n) < 1e-12
    assert abs(got["test_precision"] - prec / n) < 1e-12 and abs(got["test_recall"] - rec / n) < 1e-12
    assert abs(got["test_f1score"] - f1 / n) < 1e-12
    assert np.allclose(got["test_ious"], np.round(ious / n, 5))


def _prepared_oracle_params(version, seed, x, gt, margin, scale):
    """oracle/make_golden.py::prepare_model on the functional oracle (the recipe test_whole_model uses)."""
    sd = _model_params("vqreptunet1x1", seed)
    ks = (0, 0, 512, 512, 512)
    kw0 = dict(percent=80.0) if version == 1 else dict(th=0.7)
    with torch.no_grad():
        R.resnet_encoder(sd, x, True, momentum=1.0)
        feats = R.resnet_encoder(sd, x, False)[1:]
        for i in (2, 3, 4):
            sd[f"codebook.{i}.codebook.embedding.weight"] = cases.codebook_from_rows(cases.rows_of(feats[i]), 512, 900 + i)
        R.vq_unet_forward(sd, x, True, ks, gt=gt, version=version, margin=margin, scale=scale, momentum=1.0, **kw0)
    return sd


@pytest.mark.parametrize("version", [1, 2])
def test_cps_iterations(version):
    """SURVEY 8c fixture (9): oracle/cps_ref.py against two iterations of the trainers' loop bodies run on the reference's own
    modules (v1: with backward + Adam; v2: forward terms -- the reference's v2 backward raises, q10)."""
    from oracle.cps_ref import CPSReference
    from tests import cps_loop
    fx = golden_io.load(f"cps_iter_v{version}")
    data = cps_loop.batches(2)
    cfg = cps_loop.model_cfg(version)["params"]
    sds = [_prepared_oracle_params(version, seed, data[0][0], data[0][1], cfg["margin"], cfg["scale"]) for seed in cps_loop.SEEDS]
    ref = CPSReference(sds, version=version, margin=cfg["margin"], scale=cfg["scale"], lr=cps_loop.TRAIN["learning_rate"],
                       min_lr=cps_loop.TRAIN["min_lr"], total_iters=1000, th=cps_loop.TRAIN["confidence_threshold"],
                       drop_percent=cps_loop.TRAIN["unsup_loss_drop_percent"], proto_w=cps_loop.TRAIN["total_prototype_loss_weight"])
    for i, (l_in, l_tg, ul_in) in enumerate(data):
        out = ref.step(l_in, l_tg, ul_in, backward=fx.meta["backward"])
        for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss", "step_miou"):
            assert out[key] == pytest.approx(float(fx[f"it{i}/{key}"]), rel=2e-5), (i, key)
        assert out["lr"] == pytest.approx(float(fx[f"it{i}/lr"]), rel=1e-12)
        for key in ("mask_1", "mask_2"):                       # pseudo-label masks: exact (same ATen ops on both sides)
            assert torch.equal(out[key].to(torch.uint8), fx[f"it{i}/{key}"]), (i, key)
        for key in ("score_1", "pred_sup_1", "pred_ul_2"):
            close(out[key], fx[f"it{i}/{key}"], rtol=1e-3, atol=1e-4)
    if fx.meta["backward"]:
        for tag, p in (("m1", ref.p[0]), ("m2", ref.p[1])):
            for key in cps_loop.PROBES:
                want = fx[f"param/{tag}/{key}"]
                got = golden_io.probe(p[key])
                assert (got - want).abs().max().item() <= 1e-5 * (want.abs().max().item() + 1e-12), (tag, key)   # 0.1 x one Adam step
            close(p["encoder.bn1.running_var"], fx[f"param/{tag}/encoder.bn1.running_var"], rtol=1e-5)
            for key in cps_loop.PROBES:                        # gradients of the last iteration (still on the tensors)
                want = fx[f"it1/grad/{tag}/{key}"]
                got = golden_io.probe(p[key].grad)
                assert ((got - want).norm() / (want.norm() + 1e-30)).item() <= 2e-3, (tag, key)
        none = set(fx["it0/grad_none/m1"].tolist())
        assert {f"codebook.{i}.codebook.embedding.weight" for i in (2, 3, 4)} | {"prototype_loss.embedding.weight"} == none


def test_decoder_block_at_bench_scale():
    """oracle conv3x3_bn_relu x 2 == the reference's double_conv_block(2048, 1024) on 32 x 16 x 16 pixels (probes + checksums)."""
    fx = golden_io.load("decoder_block0_b32")
    x, sd, g = cases.block_inputs(fx.meta)
    assert synth.checksum(x) == fx.meta["x_sum"]
    p = {"blk." + k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        y = R.conv3x3_bn_relu(R.conv3x3_bn_relu(x, p, "blk.0", False), p, "blk.1", False)
    close(golden_io.probe(y), fx["y_eval"], rtol=1e-4, atol=1e-5)
    assert float(y.double().pow(2).sum()) == pytest.approx(float(fx["y_eval_stats"][1]), rel=1e-5)
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = R.conv3x3_bn_relu(R.conv3x3_bn_relu(xr, p, "blk.0", True), p, "blk.1", True)
    close(golden_io.probe(y), fx["y_train"], rtol=1e-4, atol=1e-5)
    (y * g).sum().backward()
    close(golden_io.probe(xr.grad), fx["grad_x"], rtol=1e-3, atol=1e-5 * float(fx["grad_x_stats"][2]))
    close(golden_io.probe(p["blk.0.0.weight"].grad), fx["grad_w0"], rtol=1e-3, atol=1e-5 * float(fx["grad_w0_stats"][2]))
    close(p["blk.1.1.weight"].grad, fx["grad_bn_w1"], rtol=1e-3, atol=1e-4 * float(fx["grad_bn_w1"].abs().max()))
    close(p["blk.0.1.running_var"], fx["run_var0"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_oracle_argmin_at_baseline_row_counts_on_live_codebooks(tag):
    """vq_big.npz (oracle/make_golden.py::gen_vq_big): the reference's eval-mode VectorQuantizer at N = 32768 / 8192 / 2048 / 8192 /
    16384 rows on codebooks with no dead code.  The torch restatement must return the reference's indices exactly on every case;
    the C chain oracle (the arithmetic the GPU kernel reproduces bit for bit) is checked on the cases it finishes in seconds."""
    from oracle import vq_chain
    fx = golden_io.load("vq_big")
    for case in fx.meta["cases"]:
        rows, W, idx = cases.vq_big_expected(fx, case, tag)
        n, c = rows.shape
        x = rows.reshape(1, n, 1, c).permute(0, 3, 1, 2)
        with torch.no_grad():
            q, got, loss, usage = R.vq_forward(x, W, training=False)
        assert torch.equal(got.reshape(-1), idx), case["name"]
        assert float(usage) == 0.0 == case[f"{tag}_usage"]                      # every code alive
        if n * c * case["k"] <= 2 ** 31:
            cidx, _ = vq_chain.assign(rows.numpy(), W.numpy(), vq_chain.ORDER_MFMA8)
            assert np.array_equal(cidx, idx.numpy()), case["name"]
