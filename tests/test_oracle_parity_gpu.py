"""SURVEY 8(f) rows 1-3 on the GPU against the ORACLE and the reference-captured fixtures (not against this package's own
tensor-op formulations): fused loss + pseudo-label block, on-device metrics, evaluation loop.

  fixture tests/golden/losses_metrics.npz  <- oracle/make_golden.py::gen_losses on the reference's loss/ + measurement + utils
  oracle/torch_ref.py (dice_loss, score_mask, confusion_matrix, miou, cosine_lr), oracle/cps_ref.py (regularized_pseudo_label)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cps_ref
from oracle import torch_ref as R
from tests import cases, golden_io, synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_loss_block_matches_reference_fixture(layout):
    """train_vqreptunet1x1v2.py:165-181 on the fixture's logits: supervised 0.5 CE + Dice (value, gradient), the CPS block
    (score masks exact, value, both gradients) -- through the fused one-pass kernels (vqseg_dice_ce_sums_*, vqseg_softmax_stats_f)
    and through make_loss('dice_loss') + nn.CrossEntropyLoss as the reference trainer composes them."""
    from vq_seg_amd.loss import make_loss
    from vq_seg_amd.loss.dice_loss import ce_dice_loss
    from vq_seg_amd.trainer import score_mask
    fx = golden_io.load("losses_metrics")
    pred, pred2, tgt = cases.loss_inputs()

    def put(t):
        t = t.to(dev())
        return t.contiguous(memory_format=torch.channels_last) if layout == "nhwc" else t

    tg = tgt.to(dev())
    dice, ce = make_loss("dice_loss", 3, ignore_index=255), torch.nn.CrossEntropyLoss(ignore_index=255)
    for how in ("fused", "composed"):
        p1 = put(pred).clone().requires_grad_(True)
        sup = ce_dice_loss(p1, tg, 3, 0.5, None, 255) if how == "fused" else 0.5 * ce(p1, tg) + dice(p1, tg)
        assert abs(float(sup.detach()) - float(fx["sup_loss"])) <= 2e-6 * float(fx["sup_loss"]), how
        sup.backward()
        assert rel(p1.grad, fx["sup_grad"]) < 2e-5, how
        pa, pb = put(pred).clone().requires_grad_(True), put(pred2).clone().requires_grad_(True)
        fa = score_mask(pa, torch.argmax(pa, 1).long(), fx.meta["th"])
        fb = score_mask(pb, torch.argmax(pb, 1).long(), fx.meta["th"])
        assert torch.equal(fa.cpu(), fx["filt_a"]) and torch.equal(fb.cpu(), fx["filt_b"])          # masks: exact
        if how == "fused":
            cps = ce_dice_loss(pa, fb, 3, 0.5, None, 255) + ce_dice_loss(pb, fa, 3, 0.5, None, 255)
        else:
            cps = 0.5 * ce(pa, fb) + 0.5 * ce(pb, fa) + dice(pa, fb) + dice(pb, fa)
        assert abs(float(cps.detach()) - float(fx["cps_loss"])) <= 2e-6 * float(fx["cps_loss"]), how
        cps.backward()
        assert rel(pa.grad, fx["cps_grad_a"]) < 2e-5 and rel(pb.grad, fx["cps_grad_b"]) < 2e-5, how


def test_metrics_and_schedule_match_reference_fixture():
    """measurement.py:12-62 (confusion matrix exact, mIoU / per-class IoU) and utils/lr_schedulers.py:110-112."""
    from vq_seg_amd.measurement import Measurement, confusion_matrix_device, miou_device
    from vq_seg_amd.utils.lr_schedulers import CosineAnnealingLR
    fx = golden_io.load("losses_metrics")
    pred, _, tgt = cases.loss_inputs()
    for x in (pred.to(dev()), pred.to(dev()).contiguous(memory_format=torch.channels_last)):
        conf = confusion_matrix_device(x, tgt.to(dev()), 3)
        assert torch.equal(conf.cpu(), fx["conf"].long())
        miou, ious = miou_device(conf)
        assert abs(float(miou) - float(fx["miou"])) < 1e-12 and np.allclose(ious.cpu().numpy(), fx["ious"].numpy(), rtol=1e-12)
    m = Measurement(3)                                               # the numpy surface the reference's evaluator calls
    conf_np = m._make_confusion_matrix(pred.numpy(), tgt.numpy())
    assert np.array_equal(conf_np, fx["conf"].numpy()) and abs(m.miou(conf_np)[0] - float(fx["miou"])) < 1e-12
    sched = CosineAnnealingLR(start_lr=1e-4, min_lr=1e-7, total_iters=1000, warmup_steps=0)
    assert np.allclose([sched.get_lr(i) for i in range(0, 1001, 50)], fx["lr_table"].numpy(), rtol=1e-15)


@pytest.mark.parametrize("shape,percent", [((4, 3, 64, 64), 80.0), ((2, 3, 96, 80), 83.7), ((3, 3, 33, 47), 100.0), ((8, 3, 128, 128), 90.0)])
def test_regularized_pseudo_label_matches_oracle(shape, percent):
    """make_regularized_pseudo_label (deprecated/train_with_test_pt_pseudo_entropy_reg.py:30-39): device softmax statistics +
    exact radix-select percentile against the oracle's torch ops + np.percentile on the host.  The labels agree everywhere;
    the 255 mask may differ only on pixels whose entropy is within float rounding of the threshold (it is an order statistic
    of entropies computed by two different exp/log implementations)."""
    from vq_seg_amd.trainer import regularized_pseudo_label
    logits = synth.uniform(31 + shape[0], shape, -5.0, 5.0)
    want = cps_ref.regularized_pseudo_label(logits.clone(), percent)
    got = regularized_pseudo_label(logits.to(dev()), percent).cpu()
    prob = torch.softmax(logits.double(), dim=1)
    ent = -(prob * torch.log(prob + 1e-10)).sum(1)
    thr = np.percentile(ent.numpy().flatten(), percent)
    differ = got != want
    assert int(differ.sum()) <= 2, int(differ.sum())
    assert ((ent[differ] - thr).abs() <= 1e-5 * max(thr, 1e-3)).all()
    keep = (got != 255) & (want != 255)
    assert torch.equal(got[keep], want[keep])
    assert abs(int((got == 255).sum()) - int((want == 255).sum())) <= 2


@pytest.mark.parametrize("layout", ["nchw", "nhwc", "cat"])
def test_dice_and_ce_dice_match_oracle(layout):
    """loss/dice_loss.py:5-57 (3 classes, ignore_index 255 -> zeroed logits, class-0 target) as restated in oracle/torch_ref.py,
    and 0.5 CE + Dice; value and gradient, with ignored pixels and a fully ignored image."""
    from vq_seg_amd.loss.dice_loss import ce_dice_loss, dice_loss
    b, c, h, w = 4, 3, 33, 47
    x = synth.uniform(11, (b, c, h, w), -3, 3)
    t = (synth.uniform(12, (b, h, w), 0, 1) * 3).long().clamp_(0, 2)
    t[synth.uniform(13, (b, h, w), 0, 1) > 0.8] = 255
    t[3] = 255
    xc = x.clone().requires_grad_(True)
    want = R.dice_loss(xc, t)
    (want * 2.5).backward()
    xg = x.to(dev())
    if layout == "nhwc":
        xg = xg.contiguous(memory_format=torch.channels_last)
    elif layout == "cat":
        xg = torch.cat([xg[:2].contiguous(memory_format=torch.channels_last), xg[2:]], dim=0)
    xx = xg.clone().requires_grad_(True)
    got = dice_loss(xx, t.to(dev()), 3, ignore_index=255)
    (got * 2.5).backward()
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6 and rel(xx.grad, xc.grad) < 2e-4
    xc = x.clone().requires_grad_(True)
    want = 0.5 * F.cross_entropy(xc, t, ignore_index=255) + R.dice_loss(xc, t)
    want.backward()
    xx = xg.clone().requires_grad_(True)
    got = ce_dice_loss(xx, t.to(dev()), 3, 0.5, None, 255)
    got.backward()
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-5 * abs(float(want.detach())) and rel(xx.grad, xc.grad) < 2e-5
    assert (xx.grad[3] == 0).all()


@pytest.mark.parametrize("shape", [(3, 3, 70, 61), (2, 4, 128, 96), (1, 2, 9, 5)])
def test_confusion_counts_match_oracle(shape):
    """vqseg_confusion_counts_f against oracle/torch_ref.py::confusion_matrix (Measurement._make_confusion_matrix restated), with
    arg-max ties (first maximum wins on both sides)."""
    from vq_seg_amd.measurement import confusion_matrix_device, miou_device
    b, c, h, w = shape
    logits = (synth.uniform(b * h + w, shape, -2, 2) * 4).round() / 4
    target = (synth.uniform(7, (b, h, w), 0, 1) * c).long().clamp_(0, c - 1)
    want = R.confusion_matrix(logits.numpy(), target.numpy(), c)
    got = confusion_matrix_device(logits.to(dev()).contiguous(memory_format=torch.channels_last), target.to(dev()), c)
    assert np.array_equal(got.cpu().numpy(), want)
    m, ious = miou_device(got)
    wm, wious = R.miou(want)
    assert abs(float(m) - wm) < 1e-12 and np.allclose(ious.cpu().numpy(), wious, rtol=1e-12)


def test_evaluation_loop_matches_oracle_metrics_on_reference_logits():
    """evaluate.test_loop (test_detailviz.py:87-163) on the HIP model against the metrics the ORACLE computes from the
    REFERENCE's own eval logits (fixture model_v1.npz `eval_logits`): bilinear resize to the native mask size on the CPU,
    torch_ref.confusion_matrix / miou.  Pixels whose two best classes tie within the logit tolerance may land on either
    side: at most 3 of the 2 x 96 x 80 may differ, which bounds the metric differences."""
    from tests.test_model_gpu import build
    from vq_seg_amd.evaluate import test_loop
    fx = golden_io.load("model_v1")
    model = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"])
    x, gt, _ = cases.model_inputs()
    tgt = F.interpolate(gt[:, None].float(), size=(96, 80), mode="nearest")[:, 0].long()
    got = test_loop(model, [(x, tgt)], 3, device=dev())
    ref_pred = F.interpolate(fx["eval_logits"].float(), tgt.shape[-2:], mode="bilinear")
    conf = R.confusion_matrix(ref_pred.numpy(), tgt.numpy())
    want_miou, want_ious = R.miou(conf)
    want_acc = float((ref_pred.argmax(1) == tgt).flatten(1).double().mean(1).mean())
    n_pix = tgt[0].numel()
    assert abs(got["test_miou"] - want_miou) <= 3 * 3.0 / n_pix + 1e-9, (got["test_miou"], want_miou)
    assert abs(got["test_acc"] - want_acc) <= 3.0 / n_pix + 1e-9
    assert np.allclose(got["test_ious"], np.round(want_ious, 5), atol=3 * 3.0 / n_pix + 1e-5)
