"""Round-4 step-level kernels through the C ABI:
  * vqseg_adam_step_f32 / optim.HipAdam against torch.optim.Adam (non-fused, single-tensor) on the CPU in fp32 -- the reference's
    optimiser call (train_vqreptunet1x1v2.py:106-107, :200-201): parameters and both moments within 1e-7 of scale after several
    steps; the weight images written in the same pass bit-identical to vqseg_conv_pack_all_f32 of the updated weight;
  * vqseg_conv2d_wgrad2_f (one weight-gradient launch over the two uses of a layer in a training step) against fp64 on the same
    bf16 values, for every weight-gradient kernel family."""
import copy

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from tests import synth

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


ADAM_SHAPES = [(64, 32, 3, 3), (40, 24, 3, 3), (256, 64, 1, 1), (96, 200, 1, 1), (128, 192, 3, 3), (64,), (3,), (5001,), (3, 32), (64, 3, 7, 7),
               (32, 160, 1, 1)]


def _params(device, with_kinds):
    ps = []
    for i, shp in enumerate(ADAM_SHAPES):
        p = nn.Parameter(synth.uniform(100 + i, shp, -0.2, 0.2).to(device))
        if with_kinds and len(shp) == 4 and shp[2] in (1, 3):
            cin = shp[1]
            p._vq_kinds = {"fwd", "tr"} | ({("s3", 128 if cin == 192 else cin)} if cin % 32 == 0 else set())
        ps.append(p)
    return ps


def _grads(step, device):
    return [(synth.uniform(1000 + 50 * step + i, shp, -1.0, 1.0) * (10.0 ** ((i % 5) - 3))).to(device) for i, shp in enumerate(ADAM_SHAPES)]


def test_hip_adam_matches_torch_adam_on_the_cpu_and_rewrites_the_weight_images():
    from vq_seg_amd import nnf
    from vq_seg_amd.optim import HipAdam
    cpu_p, gpu_p = _params("cpu", False), _params(dev(), True)
    # gradient storage as the trainer's buckets hand it out: views into ONE flat buffer at offsets that are not 16-byte multiples
    flat = torch.zeros(sum(p.numel() for p in gpu_p) + 1, device=dev())
    off = 1
    for p in gpu_p:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    ref = torch.optim.Adam(cpu_p, lr=3e-3, betas=(0.9, 0.999), foreach=False, fused=False)
    opt = HipAdam(gpu_p, lr=3e-3, betas=(0.9, 0.999))
    for step in range(6):
        lr = 3e-3 * (1.0 - 0.1 * step)
        for o in (ref, opt):
            o.param_groups[0]["lr"] = lr
        for p, q, g in zip(cpu_p, gpu_p, _grads(step, "cpu")):
            p.grad = g.clone()
            q.grad.copy_(g)
        ref.step()
        opt.step()
        torch.cuda.synchronize()
        for i, (p, q) in enumerate(zip(cpu_p, gpu_p)):
            for a, b, what in ((q, p, "param"), (opt.state[q]["exp_avg"], ref.state[p]["exp_avg"], "exp_avg"),
                               (opt.state[q]["exp_avg_sq"], ref.state[p]["exp_avg_sq"], "exp_avg_sq")):
                # moments: the same fma chain -> 1e-7 of scale; parameters: 2.5e-7 (<= 3 ulp at the tensor's scale) -- ATen's vectorised
                # CPU sqrt is not correctly rounded (its own scalar path differs from it by an ulp), the kernel's IEEE sqrt / div are
                assert rel(a, b) <= (2.5e-7 if what == "param" else 1e-7), (step, ADAM_SHAPES[i], what, rel(a, b))
            assert float(opt.state[q]["step"]) == float(ref.state[p]["step"]) == step + 1
    # the images installed by the step == a fresh pack of the updated weight, bit for bit; no pack launch is needed afterwards
    for q in gpu_p:
        if getattr(q, "_vq_kinds", None):
            cache = q._vq_pack
            assert "all" in cache and "fresh" not in cache              # the post-step hook consumed the mark and kept the images
            fresh = nn.Parameter(q.detach().clone())
            fresh._vq_kinds = set(q._vq_kinds)
            for kind in q._vq_kinds:
                want = nnf._pack_all(fresh, kind)
                got = nnf._pack_all(q, kind)
                if want is None:
                    assert got is None
                    continue
                assert got is cache["all"][kind], "the conv kernels must read the image the optimiser wrote"
                assert torch.equal(got, want), (tuple(q.shape), kind)
    # state_dict layout == torch.optim.Adam's: loads both ways
    plain = torch.optim.Adam(_params(dev(), False), lr=1e-3)
    plain.load_state_dict(opt.state_dict())
    opt2 = HipAdam(_params(dev(), False), lr=1e-3)
    opt2.load_state_dict(plain.state_dict())
    assert torch.equal(opt2.state[opt2.param_groups[0]["params"][0]]["exp_avg"], opt.state[gpu_p[0]]["exp_avg"])


def test_convolution_after_a_hip_adam_step_sees_the_new_weights():
    """the twin of tests/test_wcache.py::test_fused_adam_step_reaches_the_conv_kernels for optim.HipAdam"""
    from vq_seg_amd import nnf
    from vq_seg_amd.optim import HipAdam
    torch.manual_seed(0)
    for k, dtype in ((3, torch.bfloat16), (1, torch.bfloat16), (3, torch.float32)):
        conv, bn = nn.Conv2d(32, 64, k, padding=k // 2, bias=False).to(dev()), nn.BatchNorm2d(64).to(dev())
        x = torch.rand(2, 32, 16, 16, device=dev()).contiguous(memory_format=torch.channels_last).to(dtype).requires_grad_(True)
        opt = HipAdam(list(conv.parameters()) + list(bn.parameters()), lr=0.05)
        for _ in range(2):
            y0 = nnf.conv_bn_act(x, conv, bn)
            y0.float().square().mean().backward()
            opt.step()
            opt.zero_grad()
        y1 = nnf.conv_bn_act(x, conv, bn)
        fresh_c, bn2 = copy.deepcopy(conv), copy.deepcopy(bn)
        fresh_c.weight._vq_pack = None
        assert torch.equal(y1, nnf.conv_bn_act(x, fresh_c, bn2)), (k, dtype)
        assert not torch.equal(y1, y0)


def test_hip_adam_skips_parameters_without_a_gradient_and_refuses_other_adam_variants():
    from vq_seg_amd.optim import HipAdam
    a, b = nn.Parameter(torch.ones(8, device=dev())), nn.Parameter(torch.ones(8, device=dev()))
    opt = HipAdam([a, b], lr=0.1)
    a.grad = torch.full_like(a, 2.0)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(b.detach().cpu(), torch.ones(8)) and b not in opt.state        # torch.optim.Adam's rule: no grad, no state
    assert torch.allclose(a.detach().cpu(), torch.full((8,), 0.9), atol=1e-6)             # first step: lr * sign(g)
    opt.param_groups[0]["amsgrad"] = True
    with pytest.raises(NotImplementedError):
        opt.step()


# ------------------------------------------------------------------------------------------------ two-use weight gradient
PAIR_CASES = [
    # k, stride, pad, reflect, n_a, n_b, c1, c2, cout, h, w, precise
    (3, 1, 1, False, 2, 2, 64, 0, 128, 8, 32, 0),           # nine-tap kernel
    (3, 1, 1, True, 1, 3, 128, 64, 64, 12, 16, 0),          # ... concat, reflect, different image counts
    (3, 1, 1, False, 2, 1, 32, 0, 32, 16, 48, 0),
    (1, 1, 0, False, 2, 2, 128, 0, 256, 16, 16, 0),         # 1x1 kernel
    (1, 2, 0, False, 1, 2, 256, 0, 512, 15, 9, 0),          # ... stride 2, ragged
    (1, 1, 0, False, 3, 2, 64, 0, 256, 9, 7, 0),
    (3, 2, 1, True, 2, 2, 128, 0, 128, 16, 16, 0),          # per-tap kernel (stride-2 3x3)
    (3, 1, 1, False, 2, 3, 24, 16, 40, 9, 11, 0),           # ... odd channel counts, concat
    (3, 1, 1, False, 1, 2, 32, 0, 64, 8, 8, 1),             # ... precise (fp32 activations)
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_wgrad_over_two_uses_in_one_launch(case):
    from vq_seg_amd import _hip
    k, stride, pad, reflect, na, nb, c1, c2, cout, h, w, precise = case
    cin = c1 + c2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    L = _hip.lib()
    dt = torch.float32 if precise else torch.bfloat16
    seed = sum(case)
    xs = [synth.uniform(seed + i, (n, h, w, cin), -1, 1).to(dt) for i, n in enumerate((na, nb))]
    gys = [synth.uniform(seed + 7 + i, (n, ho, wo, cout), -1, 1).to(dt) for i, n in enumerate((na, nb))]
    ref = 0
    for x, gy in zip(xs, gys):
        xp = x.double().permute(0, 3, 1, 2)
        if pad:
            xp = F.pad(xp, (pad,) * 4, mode="reflect" if reflect else "constant")
        ref = ref + torch.nn.grad.conv2d_weight(xp, (cout, cin, k, k), gy.double().permute(0, 3, 1, 2), stride=stride)
    d = dev()
    xa = [x[..., :c1].contiguous().to(d) for x in xs]
    xb = [x[..., c1:].contiguous().to(d) if c2 else None for x in xs]
    gd = [g.to(d) for g in gys]
    ptr = lambda t: t.data_ptr() if t is not None else None
    nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(na + nb, h, w, cin, ho, wo, cout, k, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    gw = torch.full((cout, cin, k, k), float("nan"), dtype=torch.float32, device=d)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.vqseg_conv2d_wgrad2_f(ptr(gd[0]), ptr(xa[0]), ptr(xb[0]), na, ptr(gd[1]), ptr(xa[1]), ptr(xb[1]), nb, c1, h, w, cin, ho, wo, cout, k, k,
                                 stride, pad, int(reflect), precise, cin, 0, 0, ws.data_ptr(), nbytes, gw.data_ptr(), st)
    assert rc == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    tol = 2e-4 if precise else 2e-5
    assert rel(gw, ref) < tol
    # == the two single-use launches accumulated (different slab partition: rounding-level agreement)
    gw2 = torch.zeros_like(gw)
    for i, n in enumerate((na, nb)):
        nb1 = L.vqseg_conv2d_wgrad_workspace_bytes(n, h, w, cin, ho, wo, cout, k, k)
        ws1 = torch.empty(nb1, dtype=torch.uint8, device=d)
        rc = L.vqseg_conv2d_wgrad_f(ptr(gd[i]), ptr(xa[i]), ptr(xb[i]), c1, n, h, w, cin, ho, wo, cout, k, k, stride, pad, int(reflect), precise,
                                    cin, 0, 1, ws1.data_ptr(), nb1, gw2.data_ptr(), st)
        assert rc == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    assert rel(gw, gw2) < 2e-6
    # a second source without its tensors is refused
    assert L.vqseg_conv2d_wgrad2_f(ptr(gd[0]), ptr(xa[0]), ptr(xb[0]), na, None, None, None, nb, c1, h, w, cin, ho, wo, cout, k, k, stride, pad,
                                   int(reflect), precise, cin, 0, 0, ws.data_ptr(), nbytes, gw.data_ptr(), st) != 0


@pytest.mark.parametrize("mode", ["fast", "precise"])
def test_two_use_weight_gradients_pair_up_in_backward_and_nothing_stays_queued(mode):
    """nnf: with gradient sinks a layer used twice in one graph queues its first use and issues ONE launch at the second;
    results agree with the unpaired path, the callback fires once, a use that never meets a partner is flushed."""
    from vq_seg_amd import nnf, _hip
    dt = torch.float32 if mode == "precise" else torch.bfloat16
    torch.manual_seed(0)
    convs = [nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.Conv2d(64, 128, 1, 1, 0, bias=False), nn.Conv2d(64, 64, 3, 2, 1, bias=False, padding_mode="reflect")]
    xa = synth.uniform(1, (2, 64, 16, 16), -1, 1).to(dev()).contiguous(memory_format=torch.channels_last).to(dt)
    xb = synth.uniform(2, (3, 64, 16, 16), -1, 1).to(dev()).contiguous(memory_format=torch.channels_last).to(dt)
    for conv in convs:
        conv = conv.to(dev())
        bn = nn.BatchNorm2d(conv.out_channels).to(dev())
        results = []
        for pair in (1, 0):
            c2, b2 = copy.deepcopy(conv), copy.deepcopy(bn)
            fired = []
            for p_ in list(c2.parameters()) + list(b2.parameters()):
                p_.grad = torch.zeros_like(p_)
                p_._vq_grad_sink = lambda q: fired.append(id(q))
            _hip.PY_OPTS["py_wgrad_pair"] = pair
            try:
                (nnf.conv_bn_act(xa, c2, b2).float().square().sum() + nnf.conv_bn_act(xb, c2, b2).float().sum()).backward()
                assert not nnf._PENDING_WGRADS and getattr(c2.weight, "_vq_wgrad_pending", None) is None
            finally:
                _hip.PY_OPTS.pop("py_wgrad_pair", None)
            assert sorted(fired) == sorted(id(q) for q in list(c2.parameters()) + list(b2.parameters()))
            results.append(c2.weight.grad.clone())
        assert rel(results[0], results[1]) < (1e-5 if mode == "precise" else 1e-5)
        # a queued use whose partner never comes: flushed on demand, same gradient as the direct launch
        c3, b3 = copy.deepcopy(conv), copy.deepcopy(bn)
        for p_ in list(c3.parameters()) + list(b3.parameters()):
            p_.grad = torch.zeros_like(p_)
            p_._vq_grad_sink = None
        ya, yb = nnf.conv_bn_act(xa, c3, b3), nnf.conv_bn_act(xb, c3, b3)          # two uses counted, only one reaches the loss
        ya.float().square().sum().backward()
        assert c3.weight in nnf._PENDING_WGRADS and float(c3.weight.grad.abs().sum()) == 0.0
        assert nnf.flush_pending_wgrads() == 1 and not nnf._PENDING_WGRADS
        c4, b4 = copy.deepcopy(conv), copy.deepcopy(bn)
        nnf.conv_bn_act(xa, c4, b4).float().square().sum().backward()
        assert rel(c3.weight.grad, c4.weight.grad) < 1e-6
        del yb


# ------------------------------------------------------------------------------------------------ fan-in fusion
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fanin_kernels_are_bit_identical_to_a_separate_add(dtype):
    """The three kernels that absorb a second consumer's gradient (stride-2 1x1 data gradient in place, max-pool backward, head
    backward) against the same kernels followed by autograd's add of the two tensors."""
    from vq_seg_amd import _hip, nnf
    L = _hip.lib()
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    bf = int(dtype == torch.bfloat16)
    # -- stride-2 1x1 projection: gx (n, h, w, cin) += data gradient at the even pixels
    n, h, w, cin, cout = 2, 15, 9, 64, 128
    ho, wo = (h + 1) // 2, (w + 1) // 2
    wt = nn.Parameter(synth.uniform(5, (cout, cin, 1, 1), -0.3, 0.3).to(d))
    s_hi, s_lo = nnf._s2_weights(wt, dtype == torch.float32)
    gy = synth.uniform(6, (n, ho, wo, cout), -1, 1).to(d).to(dtype)
    other = synth.uniform(7, (n, h, w, cin), -1, 1).to(d).to(dtype)
    ptr = lambda t: t.data_ptr() if t is not None else None
    plain = torch.empty((n, h, w, cin), dtype=dtype, device=d)
    assert L.vqseg_conv2d_dgrad_s2_f(ptr(gy), ptr(s_hi), ptr(s_lo), ptr(plain), n, ho, wo, cout, cin, 1, h, w, 1 - bf, 0, st) == 0, L.vqseg_last_error()
    acc = other.clone()
    assert L.vqseg_conv2d_dgrad_s2_f(ptr(gy), ptr(s_hi), ptr(s_lo), ptr(acc), n, ho, wo, cout, cin, 1, h, w, 1 - bf, 1, st) == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    assert torch.equal(acc, plain + other)
    assert L.vqseg_conv2d_dgrad_s2_f(ptr(gy), ptr(s_hi), ptr(s_lo), ptr(acc), n, ho, wo, cout, cin, 3, h + 2, w + 2, 1 - bf, 1, st) != 0     # k = 3: refused
    # -- max-pool backward with an addend
    x = synth.uniform(8, (2, 64, 16, 20), -1, 1).to(d).contiguous(memory_format=torch.channels_last).to(dtype).requires_grad_(True)
    y = nnf.max_pool_3x3_s2(x)
    g = synth.uniform(9, tuple(y.shape), -1, 1).to(d).contiguous(memory_format=torch.channels_last).to(dtype)
    (gx_plain,) = torch.autograd.grad(y, x, g, retain_graph=True)
    add = synth.uniform(10, (2, 16, 20, 64), -1, 1).to(d).to(dtype)
    prev = nnf.set_fanin_fusion(True)
    try:
        x2 = x.detach().clone().requires_grad_(True)
        nnf.fanin_tag(x2)
        y2 = nnf.max_pool_3x3_s2(x2)
        assert nnf._fanin_deposit(x2._vq_fanin, add)
        (gx_add,) = torch.autograd.grad(y2, x2, g)
        nnf.check_fanin_consumed()
    finally:
        nnf.set_fanin_fusion(prev)
    assert torch.equal(gx_add, gx_plain + add.permute(0, 3, 1, 2))
    # -- head backward with an addend (the prototype loss's gradient of the decoder output)
    xh = synth.uniform(11, (2, 32, 8, 8), -1, 1).to(d).contiguous(memory_format=torch.channels_last).to(dtype).requires_grad_(True)
    wh = nn.Parameter(synth.uniform(12, (3, 32, 1, 1), -0.5, 0.5).to(d))
    gl = synth.uniform(13, (2, 3, 8, 8), -1, 1).to(d)
    (g_plain,) = torch.autograd.grad(nnf.head_conv1x1(xh, wh), xh, gl)
    addh = synth.uniform(14, (2, 8, 8, 32), -1, 1).to(d).to(dtype)
    prev = nnf.set_fanin_fusion(True)
    try:
        xh2 = xh.detach().clone().requires_grad_(True)
        nnf.fanin_tag(xh2)
        out = nnf.head_conv1x1(xh2, wh)
        assert nnf._fanin_deposit(xh2._vq_fanin, addh)
        (g_add,) = torch.autograd.grad(out, xh2, gl)
        nnf.check_fanin_consumed()
        # a deposit nobody absorbs is reported, not lost silently
        xh3 = xh.detach().clone().requires_grad_(True)
        nnf.fanin_tag(xh3)
        assert nnf._fanin_deposit(xh3._vq_fanin, addh)
        with pytest.raises(RuntimeError, match="never absorbed"):
            nnf.check_fanin_consumed()
    finally:
        nnf.set_fanin_fusion(prev)
    assert torch.equal(g_add, g_plain + addh.permute(0, 3, 1, 2))


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_trainer_step_with_fanin_fusion_matches_the_unfused_step(amp):
    """CPSTrainer with the five fan-in adds of a backward pass inside the consumers' kernels vs autograd's separate adds: the adds
    at the pool / head are bit-identical, the encoder features' three-way sums associate differently ((a + b) + c vs a + (b + c)):
    rounding-level agreement of the step's terms and of the parameters after two steps."""
    from tests import cps_loop
    from vq_seg_amd import _hip
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer
    outs = []
    for fan in (1, 0):
        _hip.PY_OPTS["py_fanin"] = fan
        try:
            cfg = CPSConfig(model=cps_loop.model_cfg(1, (0, 0, 32, 32, 32)), recipe="v1", total_iters=10, amp_dtype=amp, seed=3)
            tr = CPSTrainer(cfg, dev())
            res = []
            for i, (l_in, l_tg, ul_in) in enumerate(cps_loop.batches(2, 64, 2)):
                o = tr.step(l_in.to(dev()), l_tg.to(dev()), ul_in.to(dev()))
                res.append({k: float(v) for k, v in o.items()})
            probe = torch.cat([p.detach().float().reshape(-1)[:64] for p in tr.models[0].parameters()]).cpu()
            outs.append((res, probe))
        finally:
            _hip.PY_OPTS.pop("py_fanin", None)
    tol = 2e-2 if amp is not None else 2e-4
    for a, b in zip(outs[0][0], outs[1][0]):
        for k in ("loss", "sup_loss_1", "cps_loss", "commitment_loss", "prototype_loss"):
            assert abs(a[k] - b[k]) <= tol * abs(b[k]) + 1e-6, (k, a[k], b[k])
    assert rel(outs[0][1], outs[1][1]) < (5e-2 if amp is not None else 1e-3)


# ------------------------------------------------------------------------------------------------ reflect-padding data gradient by ring
@pytest.mark.parametrize("shape", [(2, 64, 64, 16, 16), (1, 128, 64, 12, 20), (3, 64, 128, 8, 8), (2, 256, 256, 4, 4), (1, 64, 64, 17, 13)])
def test_reflect_data_gradient_by_border_ring_matches_the_padded_gradient_path(shape):
    """3x3 / stride 1 / reflect-pad-1 convolution (every Bottleneck conv2, resnet.py:134-148): the data gradient as zero-padded
    gradient + border ring (vqseg_reflect_ring_f) against the padded-gradient + fold path it replaces and against autograd of
    F.conv2d on the reflect-padded input in float64 (same bf16 values)."""
    from vq_seg_amd import _hip, nnf
    n, cin, cout, h, w = shape
    torch.manual_seed(0)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False, padding_mode="reflect").to(dev())
    bn = nn.BatchNorm2d(cout).to(dev())
    bn.eval()                                                    # fixed affine: the comparison is about the convolution's data gradient
    x = synth.uniform(40, (n, cin, h, w), -1, 1).to(dev()).contiguous(memory_format=torch.channels_last).bfloat16()
    g = synth.uniform(41, (n, cout, h, w), -1, 1).to(dev()).contiguous(memory_format=torch.channels_last).bfloat16()
    outs = []
    for ring in (1, 0):
        _hip.PY_OPTS["py_reflect_ring"] = ring
        try:
            xi = x.clone().requires_grad_(True)
            y = nnf.conv_bn_act(xi, conv, bn, relu=False)
            (gx,) = torch.autograd.grad(y, xi, g)
        finally:
            _hip.PY_OPTS.pop("py_reflect_ring", None)
        outs.append(gx.float())
    xd = x.double().cpu().requires_grad_(True)
    wq = conv.weight.detach().bfloat16().double().cpu()
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().double().cpu()
    yd = F.conv2d(F.pad(xd, (1, 1, 1, 1), mode="reflect"), wq) * scale[None, :, None, None]
    (ref,) = torch.autograd.grad(yd, xd, g.double().cpu())
    assert rel(outs[0], ref) < 2 ** -7 and rel(outs[1], ref) < 2 ** -7          # bf16 outputs of bf16 operands
    assert rel(outs[0], outs[1]) < 2 ** -7                                        # border pixels: two bf16 roundings in either path
    if h > 4 and w > 4:
        inner = (slice(None), slice(None), slice(2, h - 2), slice(2, w - 2))      # interior: the same zero-padded correlation
        assert rel(outs[0][inner], outs[1][inner]) < 2 ** -8


# ------------------------------------------------------------------------------------------------ stride-2 3x3 data gradient, one launch
@pytest.mark.parametrize("case", [
    # n, channels of g_y (the layer's Cout), channels of g_x (its Cin), h, w, reflect
    (2, 128, 128, 16, 24, True), (1, 64, 64, 8, 8, False), (3, 128, 256, 4, 36, True), (2, 256, 128, 12, 8, False), (2, 64, 72, 20, 20, True),
])
def test_stride2_data_gradient_in_one_launch_without_the_padded_grid(case):
    """vqseg_conv2d_dgrad_s2_fold_f (r4): the four parity classes of a 3x3 / stride 2 / pad 1 data gradient in ONE launch, written
    straight into the unpadded gradient; reflect padding: the padded top row / left column through a small ring onto x row 1 / column 1.
    Against (a) the fp64 autograd gradient of F.conv2d on the padded input with the same bf16-rounded operands (2^-7 of the scale:
    bf16 output) and (b) the four-launch path on the padded grid + fold / crop: bit-identical away from x row 1 / column 1 (same
    accumulation order), one extra bf16 rounding there (reflect only)."""
    from vq_seg_amd import _hip, nnf
    L = _hip.lib()
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    n, cgy, cgx, h, w, reflect = case
    ho, wo = h // 2, w // 2
    wt = nn.Parameter(synth.uniform(sum(case[:5]), (cgy, cgx, 3, 3), -0.3, 0.3).to(d))
    s_hi, s_lo = nnf._s2_weights(wt, False)
    gy = synth.uniform(9, (n, ho, wo, cgy), -1, 1).bfloat16()
    # (a) fp64 reference on the bf16-rounded weights
    x = torch.zeros(n, cgx, h, w, dtype=torch.float64, requires_grad=True)
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    F.conv2d(xp, wt.detach().cpu().bfloat16().double(), stride=2).backward(gy.double().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    gyd = gy.to(d)
    rows = int(L.vqseg_conv2d_dgrad_s2_fold_rows(n, h, w, int(reflect)))
    assert rows == n * h * w + (n * (w + 1 + h) if reflect else 0) + 1
    buf = torch.full((rows, cgx), float("nan"), dtype=torch.bfloat16, device=d)
    rc = L.vqseg_conv2d_dgrad_s2_fold_f(gyd.data_ptr(), s_hi.data_ptr(), buf.data_ptr(), n, ho, wo, cgy, cgx, h, w, int(reflect), st)
    assert rc == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    got = buf[:n * h * w].view(n, h, w, cgx).float().cpu()
    assert torch.isfinite(got).all()
    assert rel(got, ref) < 2 ** -7
    # (b) the four-launch path
    gp = torch.empty((n, h + 2, w + 2, cgx), dtype=torch.bfloat16, device=d)
    assert L.vqseg_conv2d_dgrad_s2_f(gyd.data_ptr(), s_hi.data_ptr(), None, gp.data_ptr(), n, ho, wo, cgy, cgx, 3, h + 2, w + 2, 0, 0, st) == 0, L.vqseg_last_error()
    if reflect:
        old = torch.empty((n, h, w, cgx), dtype=torch.bfloat16, device=d)
        assert L.vqseg_reflect_fold_f(1, gp.data_ptr(), n, h, w, cgx, old.data_ptr(), st) == 0, L.vqseg_last_error()
    else:
        old = gp[:, 1:h + 1, 1:w + 1, :].contiguous()
    torch.cuda.synchronize()
    old = old.float().cpu()
    keep = torch.ones(h, w, dtype=torch.bool)
    if reflect:
        keep[1, :] = False
        keep[:, 1] = False
    assert torch.equal(got[:, keep], old[:, keep])
    assert (got - old).abs().max() <= 2 ** -7 * old.abs().max()
    # the option switches the entry point off (callers then take the four-launch path)
    prev = _hip.set_option("conv_dgrad_s2_merge", 0)
    try:
        assert L.vqseg_conv2d_dgrad_s2_fold_f(gyd.data_ptr(), s_hi.data_ptr(), buf.data_ptr(), n, ho, wo, cgy, cgx, h, w, int(reflect), st) != 0
    finally:
        _hip.set_option("conv_dgrad_s2_merge", prev)
