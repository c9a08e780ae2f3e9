"""The bf16 candidate filter (csrc/vq_kernels.hip: vq_filter_bf16_kernel -> vq_resolve_kernel -> vq_rescore_kernel, the exact kernel's
fmaf chains for the few candidate codes of the rows the filter leaves open) against the exact fp32-MFMA kernel on every row and against the CPU chain oracle (oracle/vq_chain.c, order "mfma8"): the
winning index AND the bits of the winning distance must be identical -- the filter is a speed path, not an approximation.
Reference arithmetic replaced: torch.cdist -> argmin of EuclideanCodebook.forward (vector_quantizer/vq_img.py:167-168) on the bf16
activations the reference's autocast region hands to the layer."""
import numpy as np
import pytest
import torch

from tests import cases, golden_io, synth

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def both_paths(rows16, W, prepared=None):
    """(idx, dmin, candidate pairs the filter handed to the exact re-score, filter launches) with the filter, (idx, dmin) without"""
    from vq_seg_amd import _hip
    before = _hip.set_option("vq_filter_launches", 0)
    idx_f, d_f, amb = _hip.vq_assign(rows16, W, want_dmin=True, prepared=prepared, want_filter_count=True)
    torch.cuda.synchronize()
    took = _hip.set_option("vq_filter_launches", before)
    prev = _hip.set_option("vq_bf16_filter", 0)
    try:
        idx_e, d_e = _hip.vq_assign(rows16, W, want_dmin=True, prepared=prepared)
        torch.cuda.synchronize()
    finally:
        _hip.set_option("vq_bf16_filter", prev)
    return (idx_f, d_f, None if amb is None else int(amb.item()), took), (idx_e, d_e)


def assert_identical(f, e, what):
    assert torch.equal(f[0], e[0]), f"{what}: {int((f[0] != e[0]).sum())} of {f[0].numel()} indices differ between the filter and the exact kernel"
    assert np.array_equal(f[1].cpu().numpy().view(np.uint32), e[1].cpu().numpy().view(np.uint32)), f"{what}: distance bits differ"


def test_filter_equals_exact_kernel_and_chain_oracle_at_baseline_row_counts():
    """vq_big.npz's five cases ((32768, 512, 512), (8192, 1024, 512), (2048, 2048, 512), (8192, 512, 256),
    (16384, 512, 1024); live k-means codebooks of the reference), bf16-exact rows through the bf16 entry point: 0 mismatches
    against the exact kernel (indices and distance bits), the reference's indices as in test_vq_gpu, and the CPU chain oracle on a
    slice.  The share of rows the filter could not decide is printed (it is what the exact re-score costs)."""
    from oracle import vq_chain
    fx = golden_io.load("vq_big")
    report = []
    for case in fx.meta["cases"]:
        rows, W, idx_ref = cases.vq_big_expected(fx, case, "bf16")
        rd, Wd = rows.to(dev()).bfloat16(), W.to(dev())
        f, e = both_paths(rd, Wd)
        assert f[3] == 1, "the bf16 entry point did not take the candidate filter"
        assert_identical(f, e, case["name"])
        n_bad, gap, excess = cases.near_tie_audit(rows.to(dev()), Wd, f[0], idx_ref.to(dev()))
        assert n_bad <= case["bf16_n_gap_lt_1e5"] and gap < 1e-5 and excess < 1e-5
        sl = slice(0, 1024)
        ref_i, ref_d = vq_chain.assign(rows[sl].numpy(), W.numpy(), vq_chain.ORDER_MFMA8)
        assert np.array_equal(f[0][sl].cpu().numpy(), ref_i) and np.array_equal(f[1][sl].cpu().numpy().view(np.uint32), ref_d.view(np.uint32))
        report.append(f"{case['name']}: N {rows.shape[0]} C {rows.shape[1]} K {W.shape[0]}: {f[2]} candidate pairs re-scored exactly "
                      f"({f[2] / rows.shape[0]:.3f} per row); {n_bad} indices differ from the reference")
        assert f[2] < 4 * rows.shape[0], "the candidate list overflowed on a separated codebook: the filter would not pay"
    print("\n".join(report))


ADVERSARIAL = ["duplicated_codes", "midpoints", "zero_rows", "rows_are_codes", "collapsed", "nearly_collapsed", "large_magnitude", "signed", "ragged"]


@pytest.mark.parametrize("kind", ADVERSARIAL)
def test_filter_on_adversarial_inputs_equals_the_exact_kernel(kind):
    """ties, near-ties and degenerate codebooks: whatever the filter cannot decide must reach the exact kernel"""
    from oracle import vq_chain
    n, c, k = 3000, 64, 256
    rows = synth.relu_features(300, (n, c))
    W = synth.relu_features(301, (k, c), sparsity=0.3, scale=1.5)
    if kind == "duplicated_codes":
        W[k // 2:] = W[:k // 2]
    elif kind == "midpoints":
        a, b = torch.arange(n) % k, (torch.arange(n) * 7 + 3) % k
        rows = 0.5 * (W[a] + W[b])
    elif kind == "zero_rows":
        rows[::3] = 0.0
    elif kind == "rows_are_codes":
        rows = W[torch.arange(n) % k].clone()
    elif kind == "collapsed":
        W[:] = W[0]
    elif kind == "nearly_collapsed":
        W = W[0][None, :] + 1e-6 * synth.uniform(302, (k, c), -1, 1)
    elif kind == "large_magnitude":
        rows, W = rows * 3.0e4, W * 2.0e4
    elif kind == "signed":
        rows, W = synth.uniform(303, (n, c), -2, 2), synth.uniform(304, (k, c), -2, 2)
    elif kind == "ragged":
        n = 2999 - 128 + 5
        rows = rows[:n]
    rows = rows.bfloat16().float()                               # the layer's bf16 activations: exact bf16 values
    f, e = both_paths(rows.to(dev()).bfloat16(), W.to(dev()))
    assert f[3] == 1
    assert_identical(f, e, kind)
    ref_i, ref_d = vq_chain.assign(rows.numpy(), W.numpy(), vq_chain.ORDER_MFMA8)
    assert np.array_equal(f[0].cpu().numpy(), ref_i), kind
    assert np.array_equal(f[1].cpu().numpy().view(np.uint32), ref_d.view(np.uint32)), kind
    print(f"{kind}: {f[2]} candidate pairs for {rows.shape[0]} rows (more than {4 * rows.shape[0]}: list overflow -> exact kernel on every row)")


def test_shapes_the_filter_does_not_serve_fall_back_to_the_exact_kernel():
    from vq_seg_amd import _hip
    for n, c, k in ((500, 64, 96), (500, 40, 256), (500, 64, 300)):
        rows = synth.relu_features(310, (n, c)).bfloat16()
        W = synth.relu_features(311, (k, c))
        before = _hip.set_option("vq_filter_launches", 0)
        idx = _hip.vq_assign(rows.to(dev()), W.to(dev()))
        torch.cuda.synchronize()
        assert _hip.set_option("vq_filter_launches", before) == 0
        assert _hip.lib().vqseg_vq_filter_counter_offset(n, c, k) == 0
        from oracle import vq_chain
        ref_i, _ = vq_chain.assign(rows.float().numpy(), W.numpy(), vq_chain.ORDER_MFMA8)
        assert np.array_equal(idx.cpu().numpy(), ref_i)


@pytest.mark.parametrize("training", [False, True])
def test_grouped_forward_with_the_filter_is_bit_identical_to_the_exact_path(training):
    """the three levels of a model forward (one filter launch) against the same call with the filter off; and with every row
    forced through the exact kernel (vq_filter_force_all)"""
    from vq_seg_amd import _hip
    shapes = [(4096, 512, 512), (1024, 1024, 512), (256, 2048, 512)]
    rows = [synth.relu_features(320 + i, (n, c)).bfloat16().to(dev()) for i, (n, c, k) in enumerate(shapes)]
    books = [synth.relu_features(330 + i, (k, c), sparsity=0.3, scale=1.5).to(dev()) for i, (n, c, k) in enumerate(shapes)]
    preps = [_hip.vq_prepare(w) for w in books]
    outs = {}
    for mode in ("filter", "force_all", "exact"):
        prev_f = _hip.set_option("vq_bf16_filter", 0 if mode == "exact" else 1)
        prev_a = _hip.set_option("vq_filter_force_all", 1 if mode == "force_all" else 0)
        try:
            outs[mode] = _hip.vq_forward_group(rows, books, preps, training, [1.0, 0.5, 0.25])
            torch.cuda.synchronize()
        finally:
            _hip.set_option("vq_bf16_filter", prev_f)
            _hip.set_option("vq_filter_force_all", prev_a)
    for mode in ("filter", "force_all"):
        for a, b in zip(outs[mode], outs["exact"]):
            for u, v in zip(a, b):
                assert torch.equal(u, v), mode
