"""Fused prototype-loss kernels (vqseg_proto_loss_*) against the module's own tensor-op formulation of the reference
(models/modules/prototype.py), which the golden fixtures pin on the CPU: both variants, margins, scales, margin modes,
fp32 and bf16 features, with and without the entropy / confidence masks; forward value and gradients."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _run(module, x, args, fused):
    from vq_seg_amd import nnf
    saved = nnf.proto_loss_supported
    if not fused:
        nnf.proto_loss_supported = lambda *_a, **_k: False
    try:
        module.embedding.weight.grad = None
        xx = x.clone().requires_grad_(True)
        loss = module(xx, *args)
        (loss.double() * 3.0).backward()
        g_emb = None if module.embedding.weight.grad is None else module.embedding.weight.grad.clone()
        return loss.detach(), xx.grad, g_emb
    finally:
        nnf.proto_loss_supported = saved


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("margin,scale,easy", [(0.0, 1.0, True), (0.3, 8.0, True), (1.5, 1.0, False), (0.0, 4.0, False)])
@pytest.mark.parametrize("variant", [1, 2])
def test_proto_loss_matches_tensor_ops(variant, margin, scale, easy, dtype):
    from vq_seg_amd.models.modules.prototype import ReliablePrototypeLoss, ReliablePrototypeLossv2
    b, c, h, w, k = 3, 32, 20, 12, 3
    cls = ReliablePrototypeLoss if variant == 1 else ReliablePrototypeLossv2
    mod = cls(k, c, scale=scale, margin=margin, init="normal", easy_margin=easy).to(dev())
    with torch.no_grad():
        mod.embedding.weight.copy_(synth.uniform(3, (k, c), -1, 1))
    mod.train()
    x = (synth.uniform(4, (b, c, h, w), -1, 1) * 2).to(dev()).contiguous(memory_format=torch.channels_last).to(dtype)
    gt = (synth.uniform(5, (b, h, w), 0, 1) * k).long().clamp_(0, k - 1).to(dev())
    if variant == 1:
        entropy = synth.uniform(6, (b * h * w,), 0, 1).to(dev())
        cases = [(gt, 80.0, entropy), (gt, 100.0, entropy)]
    else:
        scores = synth.uniform(7, (b, k, h, w), -2, 2).to(dev())
        cases = [(gt, 0.7), (scores, 0.5)]
    for args in cases:
        l_ref, gx_ref, ge_ref = _run(mod, x, args, fused=False)
        l_hip, gx_hip, ge_hip = _run(mod, x, args, fused=True)
        assert l_hip.dtype == l_ref.dtype
        assert abs(l_hip.item() - l_ref.item()) <= 2e-6 * max(1.0, abs(l_ref.item()))
        assert gx_hip.dtype == dtype and gx_hip.shape == gx_ref.shape
        assert rel(gx_hip, gx_ref) < (1e-4 if dtype == torch.float32 else 1e-2)
        if variant == 2:
            assert ge_ref is not None and ge_hip is not None and rel(ge_hip, ge_ref) < 1e-4
        else:
            assert ge_ref is None and ge_hip is None


@pytest.mark.parametrize("c", [32, 64, 8])
def test_proto_v2_prototype_gradient_over_many_row_tiles(c):
    """More than 2048 x 256 rows: a workgroup of the backward kernel walks several 256-row tiles and folds its share of
    the (K x C) prototype gradient once (row tiles in LDS, fixed order); also the widest (64) and narrowest (8) rows."""
    from vq_seg_amd.models.modules.prototype import ReliablePrototypeLossv2
    b, h, w, k = 5, 384, 384, 3                               # 737280 rows -> 2 row tiles per workgroup, a ragged tail
    mod = ReliablePrototypeLossv2(k, c, scale=4.0, margin=0.3, init="normal").to(dev())
    with torch.no_grad():
        mod.embedding.weight.copy_(synth.uniform(3, (k, c), -1, 1))
    mod.train()
    x = (synth.uniform(4, (b, c, h, w), -1, 1) * 2).to(dev()).contiguous(memory_format=torch.channels_last).bfloat16()
    x = x[:, :, :, :383] if c == 8 else x                     # 5 * 384 * 383 rows: not a multiple of 256
    x = x.contiguous(memory_format=torch.channels_last)
    gt = (synth.uniform(5, (b, h, x.shape[-1]), 0, 1) * k).long().clamp_(0, k - 1).to(dev())
    w0 = mod.embedding.weight.detach().clone()

    def run(fused):
        with torch.no_grad():                                 # forward re-normalises the prototypes in place (prototype.py:844)
            mod.embedding.weight.copy_(w0)
        return _run(mod, x, (gt, 0.7), fused=fused)

    l_ref, gx_ref, ge_ref = run(False)
    l_hip, gx_hip, ge_hip = run(True)
    assert abs(l_hip.item() - l_ref.item()) <= 2e-6 * max(1.0, abs(l_ref.item()))
    assert rel(gx_hip, gx_ref) < 1e-2 and rel(ge_hip, ge_ref) < 1e-4
    again = run(True)
    assert torch.equal(again[2], ge_hip) and torch.equal(again[1], gx_hip)          # deterministic


@pytest.mark.parametrize("layout", ["nchw", "nhwc", "cat"])
@pytest.mark.parametrize("weighted", [False, True])
def test_dice_loss_matches_tensor_ops(layout, weighted):
    """Fused Dice sums (vqseg_dice_sums_*) against the tensor-op formulation (loss/dice_loss.py), ignore_index = 255."""
    from vq_seg_amd import nnf
    from vq_seg_amd.loss.dice_loss import dice_loss
    b, c, h, w = 4, 3, 33, 47
    x = (synth.uniform(11, (b, c, h, w), -3, 3)).to(dev())
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    elif layout == "cat":
        x = torch.cat([x[:2].contiguous(memory_format=torch.channels_last), x[2:]], dim=0)
    t = (synth.uniform(12, (b, h, w), 0, 1) * 3).long().clamp_(0, 2).to(dev())
    t[synth.uniform(13, (b, h, w), 0, 1).to(dev()) > 0.8] = 255
    wgt = torch.tensor([0.2, 0.5, 0.3]) if weighted else None
    res = []
    for fused in (False, True):
        saved = nnf.dice_sums_supported
        if not fused:
            nnf.dice_sums_supported = lambda *_a, **_k: False
        try:
            xx = x.clone().requires_grad_(True)
            loss = dice_loss(xx, t, 3, weight=wgt, ignore_index=255)
            (loss * 2.5).backward()
            res.append((loss.detach(), xx.grad))
        finally:
            nnf.dice_sums_supported = saved
    assert abs(res[0][0].item() - res[1][0].item()) < 2e-6
    assert rel(res[1][1], res[0][1]) < 2e-4


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_softmax_stats_matches_tensor_ops(layout):
    """vqseg_softmax_stats_f: arg-max label, entropy and top probability of softmax(logits) in one pass."""
    from vq_seg_amd import nnf
    x = (synth.uniform(21, (3, 3, 37, 29), -6, 6)).to(dev())
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    label, ent, top = nnf.softmax_stats(x, want_top=True)
    prob = torch.softmax(x, dim=1)
    assert torch.equal(label, prob.argmax(dim=1))
    assert rel(ent, -(prob * torch.log(prob + 1e-10)).sum(dim=1)) < 1e-5
    assert rel(top, prob.max(dim=1)[0]) < 1e-6


@pytest.mark.parametrize("n,percent", [(1, 80.0), (2, 50.0), (7, 100.0), (1001, 0.0), (4099, 83.7), (1 << 20, 80.0),
                                       ((1 << 20) + 3, 99.99), (5_000_001, 92.5)])
@pytest.mark.parametrize("kind", ["entropy", "ties", "signed"])
def test_percentile_is_exact_radix_select(n, percent, kind):
    """vqseg_order_stats_f: the two order statistics around np.percentile's virtual index are EXACT (equal to the sorted
    array's entries, for clustered entropies, heavy ties, negative values and -0/+0), and the interpolated threshold masks
    the same elements as np.percentile on the host (make_regularized_pseudo_label :35-38) except, at most, elements that
    EQUAL one of the two statistics (rounding of the interpolation).  The numpy side is evaluated on the values widened to
    double: on fp32 input numpy >= 2 also rounds the QUANTILE to fp32 (q / float32(100)), which moves the virtual index
    by up to n * 2^-24 positions -- a property of the numpy version, not of the reference's algorithm."""
    import numpy as np
    from vq_seg_amd import _hip, nnf
    g = torch.Generator().manual_seed(n + int(percent * 10))
    if kind == "entropy":                               # softmax entropies of 3 classes: clustered near 0 and near log 3
        p = torch.softmax(torch.randn(n, 3, generator=g) * 4, dim=1)
        x = -(p * torch.log(p + 1e-10)).sum(1)
    elif kind == "ties":
        x = torch.randint(0, 5, (n,), generator=g).float() * 0.25
    else:
        x = torch.randn(n, generator=g)
        x[::7] = 0.0
        x[3::11] = -0.0
    xs = np.sort(x.numpy())
    virtual = percent / 100.0 * (n - 1)
    k = min(int(virtual), n - 1)
    xd = x.to(dev())
    ws = torch.empty(_hip.lib().vqseg_order_stats_workspace_bytes(), dtype=torch.uint8, device=dev())
    out = torch.empty(2, device=dev())
    rc = _hip.lib().vqseg_order_stats_f(xd.data_ptr(), n, k, ws.data_ptr(), ws.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    lo, hi = out.cpu().numpy()
    assert lo == xs[k] and hi == xs[min(k + 1, n - 1)], (lo, hi, xs[k], xs[min(k + 1, n - 1)])
    thresh = nnf.percentile(xd, percent)
    ref = np.percentile(x.numpy().astype(np.float64), percent)
    mask = (xd >= thresh).cpu().numpy()
    ref_mask = x.numpy() >= ref
    differ = mask != ref_mask
    assert not (differ & (x.numpy() != lo) & (x.numpy() != hi)).any()
    if kind == "entropy":
        assert differ.sum() <= 2
    assert abs(float(thresh) - float(ref)) <= 1e-6 * max(abs(float(ref)), 1e-3)          # fp32 lerp of exact neighbours


def test_order_stats_rejects_bad_arguments():
    from vq_seg_amd import _hip
    L = _hip.lib()
    x = torch.zeros(16, device=dev())
    ws = torch.empty(L.vqseg_order_stats_workspace_bytes(), dtype=torch.uint8, device=dev())
    out = torch.empty(2, device=dev())
    assert L.vqseg_order_stats_f(x.data_ptr(), 16, 16, ws.data_ptr(), ws.numel(), out.data_ptr(), None) != 0      # k >= n
    assert L.vqseg_order_stats_f(x.data_ptr(), 0, 0, ws.data_ptr(), ws.numel(), out.data_ptr(), None) != 0
    assert L.vqseg_order_stats_f(x.data_ptr(), 16, 3, ws.data_ptr(), 8, out.data_ptr(), None) != 0                 # workspace
    assert L.vqseg_order_stats_f(None, 16, 3, ws.data_ptr(), ws.numel(), out.data_ptr(), None) != 0


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("shape", [(3, 3, 70, 61), (2, 4, 128, 96), (1, 2, 9, 5)])
def test_confusion_counts_match_measurement(shape, layout):
    """vqseg_confusion_counts_f against Measurement._make_confusion_matrix (measurement.py:12-20) on the host: exact,
    including ties of the arg-max (first maximum) and pixels whose label is outside [0, C) (skipped)."""
    from vq_seg_amd.measurement import Measurement, confusion_matrix_device
    b, c, h, w = shape
    logits = (synth.uniform(b * h + w, shape, -2, 2) * 4).round() / 4          # quarter steps: plenty of ties
    target = (synth.uniform(7, (b, h, w), 0, 1) * c).long().clamp_(0, c - 1)
    ref = Measurement(c)._make_confusion_matrix(logits.numpy(), target.numpy())
    x = logits.to(dev())
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    got = confusion_matrix_device(x, target.to(dev()), c)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), torch.from_numpy(ref))
    assert got.sum().item() == b * h * w
    target[:, ::3, ::2] = 255                                                  # ignore label: not counted
    got = confusion_matrix_device(x, target.to(dev()), c)
    keep = target != 255
    assert got.sum().item() == keep.sum().item()
    pred = logits.argmax(1)
    for t in range(c):
        for p_ in range(c):
            assert got[:, t, p_].sum().item() == ((target == t) & (pred == p_) & keep).sum().item()


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("weighted", [False, True])
def test_fused_ce_dice_matches_the_two_reference_losses(layout, weighted):
    """0.5 * CrossEntropyLoss(ignore_index=255) + DiceLoss (the v2 recipe's terms, train_vqreptunet1x1v2.py:165-187) from
    ONE pass over the logits (vqseg_dice_ce_sums_*) against the two tensor-op formulations, value and gradient; with
    ignored pixels, an image whose pixels are all ignored, and very confident wrong predictions (large nll)."""
    from vq_seg_amd import nnf
    from vq_seg_amd.loss.dice_loss import ce_dice_loss
    b, c, h, w = 4, 3, 70, 52
    logits = synth.uniform(11, (b, c, h, w), -6, 6)
    logits[0, :, :8] *= 8.0                                                     # |logit| up to 48: nll ~ 90
    target = (synth.uniform(12, (b, h, w), 0, 1) * c).long().clamp_(0, c - 1)
    target[:, ::5, ::3] = 255
    target[3] = 255                                                             # a fully ignored image
    wt = torch.tensor([0.2, 0.5, 0.3]) if weighted else None
    x0 = logits.to(dev())
    if layout == "nhwc":
        x0 = x0.contiguous(memory_format=torch.channels_last)
    tg = target.to(dev())

    def run(fused):
        saved = nnf.dice_sums_supported
        if not fused:
            nnf.dice_sums_supported = lambda *_a, **_k: False
        try:
            x = x0.clone().requires_grad_(True)
            loss = ce_dice_loss(x, tg, c, 0.5, wt, 255)
            (loss * 2.0).backward()
            return loss.detach(), x.grad
        finally:
            nnf.dice_sums_supported = saved

    l_ref, g_ref = run(False)
    l_hip, g_hip = run(True)
    assert abs(l_hip.item() - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    assert rel(g_hip, g_ref) < 1e-5
    assert (g_hip[3] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("with_ce", [False, True])
def test_cps_loss_combination_in_one_launch_matches_the_scalar_torch_graph(with_ce):
    """nnf.cps_loss_combine (vqseg_cps_loss_combine_f, r4): sup_1 + sup_2 + w (cps_1 + cps_2) + commitment + prototype from the Dice
    (+ CE) sums, total and the gradient with respect to EVERY input, against the same expression written with torch ops (the graph
    the trainers build: dice_loss.py:27-37, train_vqreptunet1x1v2.py:165-192).  Values 2e-7 relative (the reductions associate
    differently), gradients 1e-6 of their scale."""
    from vq_seg_amd import nnf
    d = torch.device("cuda:0")
    torch.manual_seed(3)
    c, cps_w, ce_w, com_w, pro_w = 3, 1.5, (0.5 if with_ce else 0.0), 0.25, 0.01

    def term(b):
        inter = (torch.rand(b, c, device=d) * 1000).requires_grad_(True)
        sets = (torch.rand(b, c, device=d) * 3000 + 1000).requires_grad_(True)
        if not with_ce:
            return (inter, sets)
        ce = torch.stack([torch.rand(b, device=d) * 5000, torch.randint(1000, 4000, (b,), device=d).float()], dim=1).requires_grad_(True)
        return (inter, sets, ce)

    sup, cps = [term(32), term(32)], [term(64), term(64)]
    commits = [(torch.rand(3, device=d)).requires_grad_(True) for _ in range(4)]
    protos = [(torch.rand((), device=d, dtype=torch.float64)).requires_grad_(True) for _ in range(4)]

    def scalar(t):
        dice = 1 - (2 * t[0] / (t[1] + 1e-6)).mean(dim=0).mean()
        return dice if not with_ce else ce_w * (t[2][:, 0].sum() / t[2][:, 1].sum()) + dice
    commitment = (commits[0] + commits[1] + commits[2] + commits[3]) * com_w
    prototype = (protos[0] + protos[1] + protos[2] + protos[3]) * pro_w
    ref = scalar(sup[0]) + scalar(sup[1]) + cps_w * (scalar(cps[0]) + scalar(cps[1])) + commitment.sum() + prototype.float()
    leaves = [x for t in sup + cps for x in t] + commits + protos
    gref = torch.autograd.grad(ref * 1.25, leaves)               # (an upstream factor: backward must scale)
    got = nnf.cps_loss_combine(sup, cps, cps_w, ce_w, commits, com_w, protos, pro_w)
    assert got is not None
    total, stats = got
    assert not stats.requires_grad and total.requires_grad
    ggot = torch.autograd.grad(total * 1.25, leaves)
    rel1 = lambda a, b: (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)
    assert rel1(total, ref) < 2e-7
    assert rel1(stats[0], ref) < 2e-7 and rel1(stats[1], commitment.sum()) < 2e-7 and rel1(stats[2], prototype) < 2e-7
    assert rel1(stats[3], scalar(cps[0]) + scalar(cps[1])) < 2e-7 and rel1(stats[4], scalar(sup[0])) < 2e-7 and rel1(stats[7], scalar(cps[1])) < 2e-7
    for i, (a, b) in enumerate(zip(ggot, gref)):
        assert a.shape == b.shape and a.dtype == b.dtype, i
        assert rel1(a, b) < 1e-6, i
    # inputs the kernel does not take (class count mismatch, fp64 sums): the caller's torch path
    assert nnf.cps_loss_combine([(sup[0][0].double(), sup[0][1].double())], [], 1.0, 0.0, [], 0.0, [], 0.0) is None
