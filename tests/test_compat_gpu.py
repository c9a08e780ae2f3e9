"""The reference trainer's loop body, restated once in tests/cps_loop.py, driven through the reference's TOP-LEVEL module names
(`<repo>/compat` on sys.path, fresh interpreter) on the MI355X, against what the same loop body produced on the reference's own
modules on the CPU (tests/golden/cps_iter_v{1,2}.npz, SURVEY 8c fixture (9); oracle/make_golden.py::gen_cps).

Tolerances (fp32 "precise" kernels, no autocast -- like the fixture): pseudo-label masks of iteration 0 exact up to a handful of
pixels that sit on the threshold (v1: the entropy percentile is an order statistic, two pixels whose entropies differ by less than
the cross-device rounding may swap ranks; v2: top-probability > 0.7); iteration 1 comes after an Adam step, whose first update is
lr * sign(g) -- a gradient within rounding error of zero moves its weight by 2 lr the other way -- so its masks may differ in up to
0.5 % of the pixels (the CPU oracle against ITSELF with a 1e-6 input perturbation moves 10 of 16384, tests/diagnostics/
cps_mask_sensitivity.py; measured here: 45) and its logits are held to 3e-2 of their scale instead of 1e-3 (the CPU oracle against
itself: a 1e-5 input perturbation in iteration 0 moves iteration-1 logits by 0.4-1.3 % of scale; measured here 1.2 %); losses 1e-4
relative (iteration 1: 1e-3), iteration-0 logits 1e-3 of their scale (north_star), parameters after
the two Adam steps 1e-3 of their scale, gradient probes 5e-2 relative L2 (the 2x2 / 4x4 levels normalise over 8 / 32 samples at
this size; see tests/test_model_gpu.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import cps_loop, golden_io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def drive(tmp_path, what):
    out = tmp_path / f"{what}.npz"
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "compat_driver.py"), os.path.join(ROOT, "compat"), str(out), what],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    z = np.load(out, allow_pickle=False)
    assert str(z["module_of_model"]).startswith("vq_seg_amd.models.networks")        # the flat names resolved to this repository
    return z


def compare_iterations(got, fx, backward_in_fixture, mask_slack=(4, 82)):
    report = []
    for i in range(2):
        for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
            a, b = float(got[f"it{i}/{key}"]), float(fx[f"it{i}/{key}"])
            assert abs(a - b) <= (1e-4, 1e-3)[i] * abs(b) + 1e-7, (i, key, a, b)
            report.append(f"it{i} {key}: {a:.7f} vs {b:.7f}")
        assert float(got[f"it{i}/lr"]) == pytest.approx(float(fx[f"it{i}/lr"]), rel=1e-12)
        for key in ("mask_1", "mask_2"):
            diff = int((torch.from_numpy(got[f"it{i}/{key}"]) != fx[f"it{i}/{key}"]).sum())
            assert diff <= mask_slack[i], (i, key, diff)
            report.append(f"it{i} {key}: {diff} of {fx[f'it{i}/{key}'].numel()} pixels differ")
        for key in ("score_1", "pred_sup_1", "pred_ul_2"):
            a, b = torch.from_numpy(got[f"it{i}/{key}"]).double(), fx[f"it{i}/{key}"].double()
            err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
            assert err <= (1e-3, 3e-2)[i], (i, key, err)
            report.append(f"it{i} {key}: max err {err:.2e} of scale")
        assert abs(float(got[f"it{i}/step_miou"]) - float(fx[f"it{i}/step_miou"])) <= (2e-3, 1e-2)[i]
        assert np.allclose(got[f"it{i}/code_usage"], fx[f"it{i}/code_usage"].numpy(), rtol=1e-6)      # dead-code % exact
        if backward_in_fixture:
            for tag in ("m1", "m2"):
                for key in cps_loop.PROBES:
                    a, b = torch.from_numpy(got[f"it{i}/grad/{tag}/{key}"]).double(), fx[f"it{i}/grad/{tag}/{key}"].double()
                    l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
                    assert l2 <= (5e-2, 0.3)[i], (i, tag, key, l2)     # it1: the CPU oracle against itself moves 12-13 % (1e-5 input perturbation)
                    report.append(f"it{i} grad {tag} {key}: rel L2 {l2:.2e}")
    if backward_in_fixture:
        for tag in ("m1", "m2"):
            for key in cps_loop.PROBES + ["encoder.bn1.running_var"]:
                a, b = torch.from_numpy(got[f"param/{tag}/{key}"]).double(), fx[f"param/{tag}/{key}"].double()
                # Adam's step is ~ lr * sign(g): a gradient entry within rounding error of zero may move its weight the other way,
                # 2 * lr per step -> at most 4 * lr = 4e-4 after the two steps (and 1e-3 of scale for the running statistics)
                err = (a - b).abs().max().item()
                assert err <= max(1e-3 * b.abs().max().item(), 4.2 * cps_loop.TRAIN["learning_rate"]), (tag, key, err)
        none = set(fx["it0/grad_none/m1"].tolist())
        assert set(got["it0/grad_none/m1"].tolist()) == none
    print("\n".join(report))


def test_v1_loop_body_through_top_level_names(tmp_path):
    """deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203, two iterations with backward and Adam."""
    compare_iterations(drive(tmp_path, "iter_v1"), golden_io.load("cps_iter_v1"), True)


def test_v2_loop_body_through_top_level_names(tmp_path):
    """train_vqreptunet1x1v2.py:137-211, two iterations.  The reference's own v2 backward raises on fp32, so its
    fixture holds the forward terms of two iterations WITHOUT an optimiser step in between; here the backward and the Adam
    steps run (they must: that is the trainer), so only iteration 0 is comparable term by term."""
    got, fx = drive(tmp_path, "iter_v2"), golden_io.load("cps_iter_v2")
    for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
        a, b = float(got[f"it0/{key}"]), float(fx[f"it0/{key}"])
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-7, (key, a, b)
    for key in ("mask_1", "mask_2"):
        assert int((torch.from_numpy(got[f"it0/{key}"]) != fx[f"it0/{key}"]).sum()) <= 4, key
    for key in ("score_1", "pred_sup_1", "pred_ul_2"):
        a, b = torch.from_numpy(got[f"it0/{key}"]).double(), fx[f"it0/{key}"].double()
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item(), key
    assert np.isfinite(float(got["it1/loss"]))
    # prototypes DO receive a gradient in v2 (prototype.py:844-849): only the codebooks stay without one
    assert set(got["it0/grad_none/m1"].tolist()) == {f"codebook.{i}.codebook.embedding.weight" for i in (2, 3, 4)}
    for tag in ("m1", "m2"):
        for key in cps_loop.PROBES:
            assert np.isfinite(got[f"it1/grad/{tag}/{key}"]).all()


def test_miou_parity_run(tmp_path):
    """north_star: "mIoU within +-0.2".  40 v1 iterations on the synthetic crop/weed blobs at 64x64 (CWFID cannot travel), the same
    loop body, data, initial weights and codebooks on both sides: the reference's modules on the CPU (fixture cps_curve_v1.npz)
    and this repository through the flat names on the MI355X.  Test-set mIoU of model_1 after 0 / 10 / 20 / 30 / 40 steps must
    agree within 0.2 points (0.002), and training must actually have moved it."""
    got, fx = drive(tmp_path, "curve"), golden_io.load("cps_curve_v1")
    a, b = got["test_miou"], fx["test_miou"].numpy()
    print("test mIoU  GPU:", np.round(a, 5), " reference CPU:", np.round(b, 5))
    assert a.shape == b.shape
    assert abs(a[0] - b[0]) <= 1e-4                                  # same start
    assert abs(a[-1] - b[-1]) <= 0.002, (a, b)                       # 0.2 mIoU points at the end
    assert np.abs(a - b).max() <= 0.03, (a, b)                       # and never far apart on the way (steep phase: 0.05 per step)
    assert b[-1] > b[0] + 0.1                                        # the run learns (so the comparison means something)
    # the first losses: iteration 0 sees identical weights (1e-4); every later one comes after Adam steps whose first updates are
    # ~ lr * sign(g) (lr = 1e-3 here), so a last-bit difference of ANY forward kernel (r3: the bilinear kernels' spelled-out fma order)
    # moves iteration 2 by ~1e-3 -- the CPU oracle against itself under a 1e-5 input perturbation moves more
    # (tests/diagnostics/cps_mask_sensitivity.py); the end of the curve above carries the bar that matters
    # (r4: bars at <= 2x the measured deltas -- 1e-6 / 3e-4 / 1.1e-3 on the r3 build)
    for i, tol in enumerate((1e-4, 1e-3, 2.5e-3)):
        assert abs(got["sup_loss_1"][i] - fx["sup_loss_1"].numpy()[i]) <= tol * abs(fx["sup_loss_1"].numpy()[i]), (i, got["sup_loss_1"][:3], fx["sup_loss_1"][:3])


def test_miou_parity_run_bf16_autocast(tmp_path):
    """The same run at the BENCHMARKED precision (training forwards / backwards under bf16 autocast, pseudo-label forwards fp32 --
    the reference's AMP layout with its fp16 replaced by bf16) against the reference's fp32 CPU curve: the end point has to land
    within the north_star's 0.2 mIoU points as well; on the way bf16 rounding may move a step's mIoU more than fp32 does."""
    got, fx = drive(tmp_path, "curve_bf16"), golden_io.load("cps_curve_v1")
    a, b = got["test_miou"], fx["test_miou"].numpy()
    print("test mIoU  GPU bf16:", np.round(a, 5), " reference CPU fp32:", np.round(b, 5))
    assert abs(a[0] - b[0]) <= 1e-4                                  # the first evaluation precedes any training (fp32 eval)
    assert abs(a[-1] - b[-1]) <= 0.002, (a, b)
    assert np.abs(a - b).max() <= 0.05, (a, b)


def test_trainer_main_flow_on_a_synthetic_folder(tmp_path):
    """train() of train_vqreptunet1x1v2.py as far as the hot path goes -- BaseDataset folders and loaders, make_model + init_weight
    from random init (k-means codebook / prototype initialisation in the first training forward), Adam + cosine schedule, the bf16
    autocast region -- through the flat names for two epochs: every loss finite, codebooks initialised, six iterations."""
    got = drive(tmp_path, "train_main")
    assert int(got["iters"]) == 6 and np.isfinite(got["losses"]).all() and got["initted"].all()
    assert ((got["mious"] >= 0) & (got["mious"] <= 1)).all()


def _check_amp_iterations(got, fx, version):
    """The reference trainer's LITERAL mixed-precision region (`half: true` in every shipped config): `torch.cuda.amp.autocast(
    enabled=True)` with no dtype (float16 by default) + `GradScaler` (train_vqreptunet1x1v2.py:114,151,172,199-202).  This repository's
    modules compute in bfloat16 inside ANY enabled autocast region (nnf.act_dtype), so the yardstick is the SAME two
    iterations under an explicit `torch.autocast(dtype=torch.bfloat16)` without a scaler (`bf16/...`, produced by the same child
    process): every forward term and logit of the literal region (`amp/...`) must equal it (the loss scale 65536 is a power of two:
    it moves exponents only, and bf16 has fp32's exponent range), un-scaled gradients and post-step parameters within 1e-5 of scale.
    The scaler must never see an inf (scale stays 65536: no skipped optimiser step).  Against the reference's fp32 CPU fixture only
    the loose bf16 bars apply at this size (BatchNorm over 8 samples at the deepest level amplifies bf16 rounding): iteration-0 losses
    within 5e-2 relative; the fp32 pseudo-label forward (outside autocast) keeps the 1e-3 bar."""
    rep = []
    for i in range(2):
        assert float(got[f"amp/it{i}/scale_before"]) == 65536.0 and float(got[f"amp/it{i}/scale_after"]) == 65536.0, "GradScaler skipped a step"
        for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
            a, b = float(got[f"amp/it{i}/{key}"]), float(got[f"bf16/it{i}/{key}"])
            assert np.isfinite(a) and abs(a - b) <= (1e-6, 1e-3)[i] * abs(b) + 1e-9, (i, key, a, b)
            rep.append(f"it{i} {key}: literal AMP {a:.7f}  bf16 autocast {b:.7f}")
        for key in ("mask_1", "mask_2"):
            diff = int((got[f"amp/it{i}/{key}"] != got[f"bf16/it{i}/{key}"]).sum())
            assert diff <= (0, 40)[i], (i, key, diff)
        for key in ("score_1", "pred_sup_1", "pred_ul_2"):
            a, b = got[f"amp/it{i}/{key}"].astype(np.float64), got[f"bf16/it{i}/{key}"].astype(np.float64)
            err = np.abs(a - b).max() / np.abs(b).max()
            assert err <= (0.0, 2e-2)[i], (i, key, err)
            rep.append(f"it{i} {key}: literal AMP vs bf16 autocast {err:.2e} of scale")
        for tag in ("m1", "m2"):
            for key in cps_loop.PROBES:
                a, b = got[f"amp/it{i}/grad/{tag}/{key}"].astype(np.float64), got[f"bf16/it{i}/grad/{tag}/{key}"].astype(np.float64)
                assert np.isfinite(a).all(), (i, tag, key)
                if i == 0:
                    l2 = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
                    assert l2 <= 1e-5, (tag, key, l2)
                    rep.append(f"it0 grad {tag} {key}: rel L2 {l2:.2e}")
    assert set(got["amp/it0/grad_none/m1"].tolist()) == set(got["bf16/it0/grad_none/m1"].tolist())
    for key in ("loss", "sup_loss_1", "sup_loss_2", "cps_loss", "commitment_loss", "prototype_loss"):
        a, b = float(got[f"amp/it0/{key}"]), float(fx[f"it0/{key}"])
        assert abs(a - b) <= 5e-2 * abs(b) + 1e-6, (key, a, b)
        rep.append(f"it0 {key}: {a:.6f} vs reference fp32 {b:.6f}")
    a, b = torch.from_numpy(got["amp/it0/score_1"]).double(), fx["it0/score_1"].double()
    assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item()          # the pseudo-label forward is OUTSIDE autocast: fp32 bar
    for key in ("pred_sup_1", "pred_ul_2"):
        a, b = torch.from_numpy(got[f"amp/it0/{key}"]).double(), fx[f"it0/{key}"].double()
        rep.append(f"it0 {key}: {(a - b).abs().max().item() / b.abs().max().item():.2e} of scale from the reference's fp32 logits (bf16 region)")
    print("\n".join(rep))


def test_v2_literal_amp_region_with_gradscaler(tmp_path):
    """v2 loop body, fp16-default autocast + GradScaler, two iterations, against cps_iter_v2.npz."""
    _check_amp_iterations(drive(tmp_path, "iter_v2_amp"), golden_io.load("cps_iter_v2"), 2)


def test_v1_literal_amp_region_with_gradscaler(tmp_path):
    _check_amp_iterations(drive(tmp_path, "iter_v1_amp"), golden_io.load("cps_iter_v1"), 1)


def test_miou_parity_run_literal_amp(tmp_path):
    """The 40-step mIoU run under the trainer's literal AMP region (fp16-default autocast + GradScaler): no skipped step, end point
    within the north_star's 0.2 mIoU points of the reference's fp32 CPU curve."""
    got, fx = drive(tmp_path, "curve_amp"), golden_io.load("cps_curve_v1")
    a, b = got["test_miou"], fx["test_miou"].numpy()
    print("test mIoU  GPU literal AMP:", np.round(a, 5), " reference CPU fp32:", np.round(b, 5))
    assert (got["scale"] == 65536.0).all(), got["scale"]
    assert abs(a[0] - b[0]) <= 1e-4
    assert abs(a[-1] - b[-1]) <= 0.002, (a, b)
    assert np.abs(a - b).max() <= 0.05, (a, b)


@pytest.mark.parametrize("what", ["curve128", "curve128_bf16", "curve128_amp"])
def test_miou_parity_run_128_k512_200_steps(tmp_path, what):
    """the thicker mIoU-parity run.  200 v1 iterations at 128x128 with the shipped codebook size K = 512 at the
    three levels (the 64x64 run uses K = 64), test mIoU every 25 steps; reference modules on the CPU (fixture cps_curve_v1_128.npz,
    oracle/make_golden.py curve128: 0.219 -> 0.929 -> 0.984 -> ... -> 0.998) against this repository through the flat names on the
    MI355X in fp32, under bf16 autocast (the benchmarked precision) and under the trainer's literal AMP region.  North_star:
    the end point within 0.2 mIoU points."""
    got, fx = drive(tmp_path, what), golden_io.load("cps_curve_v1_128")
    a, b = got["test_miou"], fx["test_miou"].numpy()
    print(f"test mIoU  GPU {what}:", np.round(a, 5), " reference CPU fp32:", np.round(b, 5))
    assert a.shape == b.shape == (9,)
    assert abs(a[0] - b[0]) <= 1e-4                                  # same start (fp32 evaluation before any training)
    assert abs(a[-1] - b[-1]) <= 0.002, (a, b)                       # 0.2 mIoU points at the end
    assert np.abs(a[2:] - b[2:]).max() <= 0.01, (a, b)               # from step 50 on the curves stay within one point
    assert np.abs(a - b).max() <= (0.03 if what == "curve128" else 0.06), (a, b)     # steep phase (step 25)
    assert b[-1] > 0.99 and a[-1] > 0.99
    if what == "curve128_amp":
        assert (got["scale"] == 65536.0).all(), got["scale"]
