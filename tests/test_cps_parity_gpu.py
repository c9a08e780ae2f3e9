"""`vq_seg_amd.trainer.CPSTrainer` (the product's caller of the hot path: fused losses, device-side percentile, two streams,
gradient buckets, fused Adam) against CPS iterations of the REFERENCE trainers' loop bodies captured on the CPU
(tests/golden/cps_iter_v{1,2}.npz = SURVEY 8c fixture (9), oracle/make_golden.py::gen_cps), fp32 "precise" kernels.
Tolerances as in tests/test_compat_gpu.py."""
import types

import numpy as np
import pytest
import torch

from tests import cases, cps_loop, golden_io

pytestmark = pytest.mark.gpu


def _trainer(version, two_streams, **kw):
    import vq_seg_amd.models as models
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer
    dev = torch.device("cuda:0")
    ns = types.SimpleNamespace(models=models)
    pair = cps_loop.build_pair(ns, version, dev, prepare=lambda m, x, gt, v: cases.prepare_module_model(
        m, x, gt, v, to_input=lambda t: t.contiguous(memory_format=torch.channels_last)))
    t = cps_loop.TRAIN
    cfg = CPSConfig(model=cps_loop.model_cfg(version), recipe=f"v{version}", learning_rate=t["learning_rate"], min_lr=t["min_lr"],
                    total_iters=1000, cps_loss_weight=t["cps_loss_weight"], total_commitment_loss_weight=t["total_commitment_loss_weight"],
                    total_prototype_loss_weight=t["total_prototype_loss_weight"], unsup_loss_drop_percent=t["unsup_loss_drop_percent"],
                    confidence_threshold=t["confidence_threshold"], criterion=t["criterion"], init_weights=False, amp_dtype=None,
                    two_streams=two_streams, keep_aux=True, **kw)
    return CPSTrainer(cfg, dev, models=pair), dev


@pytest.mark.parametrize("two_streams", [False, True])
def test_v1_trainer_steps_match_reference_iterations(two_streams):
    fx = golden_io.load("cps_iter_v1")
    tr, dev = _trainer(1, two_streams)
    for i, (l_in, l_tg, ul_in) in enumerate(cps_loop.batches(2)):
        out = tr.step(l_in.to(dev), l_tg.to(dev), ul_in.to(dev), epoch_frac=0.0)
        for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
            a, b = float(out[key]), float(fx[f"it{i}/{key}"])
            assert abs(a - b) <= (1e-4, 1e-3)[i] * abs(b) + 1e-7, (i, key, a, b)
        assert float(out["lr"]) == pytest.approx(float(fx[f"it{i}/lr"]), rel=1e-6)
        assert abs(float(out["miou"]) - float(fx[f"it{i}/step_miou"])) <= (2e-3, 1e-2)[i]
        for key in ("mask_1", "mask_2"):
            diff = int((tr.aux[key].cpu().to(torch.uint8) != fx[f"it{i}/{key}"]).sum())
            assert diff <= (4, 82)[i], (i, key, diff)          # see tests/test_compat_gpu.py on iteration 1
        for key in ("score_1", "pred_sup_1", "pred_ul_2"):
            a, b = tr.aux[key].double().cpu(), fx[f"it{i}/{key}"].double()
            assert (a - b).abs().max().item() <= (1e-3, 3e-2)[i] * b.abs().max().item(), (i, key)   # it1: after an Adam step
        for tag, m in (("m1", tr.models[0]), ("m2", tr.models[1])):
            named = dict(m.named_parameters())
            for key in cps_loop.PROBES:                                # p.grad = the bucket view the kernels accumulated into
                a, b = golden_io.probe(named[key].grad).double().cpu(), fx[f"it{i}/grad/{tag}/{key}"].double()
                l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
                assert l2 <= (5e-2, 0.3)[i], (i, tag, key, l2)      # it1: see tests/test_compat_gpu.py (oracle vs itself: 12-13 %)
    for tag, m in (("m1", tr.models[0]), ("m2", tr.models[1])):
        sd = m.state_dict()
        for key in cps_loop.PROBES + ["encoder.bn1.running_var"]:
            a, b = golden_io.probe(sd[key]).double().cpu(), fx[f"param/{tag}/{key}"].double()
            assert (a - b).abs().max().item() <= max(1e-3 * b.abs().max().item(), 4.2 * cps_loop.TRAIN["learning_rate"]), (tag, key)   # 2 steps x 2 lr


def test_v2_trainer_step_matches_reference_forward_terms():
    """v2: the reference's backward raises on fp32 (q10), so the fixture pins the forward terms of iteration 0; the step itself
    (backward through the out-of-place prototype loss, Adam) must run and stay finite."""
    fx = golden_io.load("cps_iter_v2")
    tr, dev = _trainer(2, True)
    data = cps_loop.batches(2)
    l_in, l_tg, ul_in = data[0]
    out = tr.step(l_in.to(dev), l_tg.to(dev), ul_in.to(dev))
    for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
        a, b = float(out[key]), float(fx[f"it0/{key}"])
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-7, (key, a, b)
    for key in ("mask_1", "mask_2"):
        assert int((tr.aux[key].cpu().to(torch.uint8) != fx[f"it0/{key}"]).sum()) <= 4, key
    for key in ("score_1", "pred_sup_1", "pred_ul_2"):
        a, b = tr.aux[key].double().cpu(), fx[f"it0/{key}"].double()
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item(), key
    l_in, l_tg, ul_in = data[1]
    out = tr.step(l_in.to(dev), l_tg.to(dev), ul_in.to(dev))
    assert np.isfinite(float(out["loss"]))


def test_bf16_training_forwards_keep_fp32_pseudo_label_passes():
    """cfg.amp_dtype = bf16 with the default eval_amp=False: the two no-grad pseudo-label forwards stay fp32 like the reference's
    (train_vqreptunet1x1v2.py:143-149 are outside autocast), so the scores are the fp32 fixture's; eval_amp=True moves them."""
    fx = golden_io.load("cps_iter_v1")
    l_in, l_tg, ul_in = cps_loop.batches(1)[0]
    errs = {}
    for eval_amp in (False, True):
        from vq_seg_amd.trainer import CPSConfig, CPSTrainer   # noqa: F401
        tr, dev = _trainer(1, True, eval_amp=eval_amp)
        tr.cfg.amp_dtype = torch.bfloat16
        tr.step(l_in.to(dev), l_tg.to(dev), ul_in.to(dev))
        a, b = tr.aux["score_1"].double().cpu(), fx["it0/score_1"].double()
        errs[eval_amp] = (a - b).abs().max().item() / b.abs().max().item()
    assert errs[False] <= 1e-3 < errs[True], errs
