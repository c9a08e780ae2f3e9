"""GPU parity of the encoder/decoder building blocks (HIP kernels through the C ABI) against plain PyTorch
fp32 references of the same ops run on the CPU (this tier's oracle for floating-point kernels).
precise mode (fp32 activations, bf16x3 split MFMA): 2e-4 of the tensor scale; fast mode (bf16): 3e-2."""
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


def cl(t):
    return t.to(dev()).contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # n, cin, cout, h, w, k, stride, pad, reflect, c2 (second input channels), residual
    (2, 64, 64, 16, 16, 3, 1, 1, False, 0, False),
    (2, 64, 128, 17, 13, 3, 1, 1, True, 0, False),          # reflect, ragged spatial size
    (1, 128, 256, 16, 16, 3, 2, 1, True, 0, False),         # stride-2 reflect (bottleneck conv2 of a stage's first block)
    (2, 256, 512, 8, 8, 1, 2, 0, False, 0, False),          # 1x1 stride-2 projection shortcut
    (2, 64, 64, 15, 19, 3, 2, 1, True, 0, False),           # stride-2 data gradient by parity classes: odd sizes, reflect
    (1, 64, 128, 14, 10, 3, 2, 1, False, 0, False),         # ... zero padding (plain UNet's encoder)
    (2, 64, 128, 9, 7, 1, 2, 0, False, 0, False),           # ... 1x1, odd sizes
    (1, 24, 40, 12, 12, 3, 2, 1, True, 0, False),           # ... channel counts that are not multiples of 32
    (2, 64, 256, 12, 12, 1, 1, 0, False, 0, True),          # conv3 + residual + relu
    (2, 128, 32, 24, 24, 3, 1, 1, False, 64, False),        # decoder concat 128 + 64 -> 32
    (1, 32, 32, 40, 40, 3, 1, 1, False, 0, False),
    (2, 24, 16, 9, 11, 3, 1, 1, False, 32, False),          # odd channel counts (fixture-style decoder)
    (3, 48, 24, 5, 5, 3, 1, 1, False, 0, False),
    # shapes that take the fused nine-tap weight-gradient kernel (Ho % 4 == 0, Wo % 16 == 0): every (co, ci) tile config
    (2, 64, 128, 8, 32, 3, 1, 1, True, 0, False),           # 128-wide co tile, 64-wide ci tile, reflect
    (1, 128, 64, 12, 16, 3, 1, 1, False, 64, False),        # concat 128 + 64, 64-wide co tile
    (2, 32, 32, 16, 48, 3, 1, 1, False, 0, False),          # 32 x 32
    (1, 96, 32, 8, 16, 3, 1, 1, False, 32, False),          # split at 96 forces the 32-wide ci tile
    (2, 64, 32, 20, 16, 3, 1, 1, True, 0, False),
    (3, 32, 128, 4, 16, 3, 1, 1, False, 0, False),
    (2, 32, 64, 8, 16, 3, 1, 1, True, 0, False),
]


@pytest.mark.parametrize("mode", ["precise", "fast"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bn_act_forward_backward(case, training, mode):
    from vq_seg_amd import nnf
    n, cin, cout, h, w, k, s, p, reflect, c2, use_res = case
    if mode == "fast" and (cin % 8 or c2 % 8 or cout % 8):
        pytest.skip("bf16 mode needs channel counts that are multiples of 8")
    seed = sum(case[:8]) + 7 * int(training)
    conv = nn.Conv2d(cin + c2, cout, k, s, p, bias=False, padding_mode="reflect" if reflect else "zeros")
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(synth.uniform(seed, tuple(conv.weight.shape), -1, 1) * (2.0 / ((cin + c2) * k * k)) ** 0.5)
        bn.weight.copy_(synth.uniform(seed + 1, (cout,), 0.5, 1.5))
        bn.bias.copy_(synth.uniform(seed + 2, (cout,), -0.3, 0.3))
        bn.running_mean.copy_(synth.uniform(seed + 3, (cout,), -0.2, 0.2))
        bn.running_var.copy_(synth.uniform(seed + 4, (cout,), 0.5, 1.5))
    conv.train(training), bn.train(training)
    x = synth.uniform(seed + 5, (n, cin, h, w), -1, 1)
    x2 = synth.uniform(seed + 6, (n, c2, h, w), -1, 1) if c2 else None
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    res = synth.uniform(seed + 7, (n, cout, ho, wo), -1, 1) if use_res else None
    g = synth.uniform(seed + 8, (n, cout, ho, wo), -1, 1)
    dt = torch.float32 if mode == "precise" else torch.bfloat16
    x, g = x.to(dt).float(), g.to(dt).float()                   # the values the kernels actually receive
    x2 = x2.to(dt).float() if c2 else None
    res = res.to(dt).float() if use_res else None
    # ReLU is discontinuous: where the pre-activation is within the kernels' rounding error of zero the mask may
    # legitimately differ from the fp32 reference, so the cotangent is zeroed there (both sides).
    with torch.no_grad():
        pre = bn(conv(torch.cat((x, x2), 1) if c2 else x)) + (res if use_res else 0)
        if training:                                            # undo the running-stat update of this probe forward
            bn.running_mean.copy_(synth.uniform(seed + 3, (cout,), -0.2, 0.2))
            bn.running_var.copy_(synth.uniform(seed + 4, (cout,), 0.5, 1.5))
            bn.num_batches_tracked.zero_()
    g = g * (pre.abs() > (1e-4 if mode == "precise" else 5e-2)).float()
    # ---- CPU fp32 reference
    xr = x.clone().requires_grad_(True)
    x2r = x2.clone().requires_grad_(True) if c2 else None
    rr = res.clone().requires_grad_(True) if use_res else None
    inp = torch.cat((xr, x2r), 1) if c2 else xr
    y = bn(conv(inp))
    if use_res:
        y = y + rr
    y = F.relu(y)
    y.backward(g)
    ref = dict(y=y.detach(), gx=xr.grad, gx2=x2r.grad if c2 else None, gr=rr.grad if use_res else None,
               gw=conv.weight.grad.clone(), gg=bn.weight.grad.clone(), gb=bn.bias.grad.clone(),
               rm=bn.running_mean.clone(), rv=bn.running_var.clone())
    # ---- HIP path
    import copy
    conv_g, bn_g = copy.deepcopy(conv).to(dev()), copy.deepcopy(bn).to(dev())
    for m_ in (conv_g, bn_g):
        for p_ in m_.parameters():
            p_.grad = None
    with torch.no_grad():
        bn_g.running_mean.copy_(synth.uniform(seed + 3, (cout,), -0.2, 0.2))
        bn_g.running_var.copy_(synth.uniform(seed + 4, (cout,), 0.5, 1.5))
    xg = cl(x).to(dt).requires_grad_(True)
    x2g = cl(x2).to(dt).requires_grad_(True) if c2 else None
    rg = cl(res).to(dt).requires_grad_(True) if use_res else None
    out = nnf.conv_bn_act(xg, conv_g, bn_g, relu=True, residual=rg, x2=x2g)
    assert out.dtype == dt and out.shape == ref["y"].shape
    out.backward(cl(g).to(dt))
    tol = 2e-4 if mode == "precise" else 3e-2
    assert rel(out.float(), ref["y"]) < tol, "forward"
    assert rel(xg.grad.float(), ref["gx"]) < tol, "grad x"
    if c2:
        assert rel(x2g.grad.float(), ref["gx2"]) < tol, "grad x2"
    if use_res:
        assert rel(rg.grad.float(), ref["gr"]) < tol, "grad residual"
    assert rel(conv_g.weight.grad, ref["gw"]) < tol, "grad weight"
    assert rel(bn_g.weight.grad, ref["gg"]) < tol, "grad gamma"
    assert rel(bn_g.bias.grad, ref["gb"]) < tol, "grad beta"
    if training:
        assert rel(bn_g.running_mean, ref["rm"]) < (1e-5 if mode == "precise" else 1e-2)
        assert rel(bn_g.running_var, ref["rv"]) < (1e-5 if mode == "precise" else 1e-2)
        assert int(bn_g.num_batches_tracked) == int(bn.num_batches_tracked) + 1   # (copied after the CPU forward)


@pytest.mark.parametrize("mode", ["precise", "fast"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bn_act_eval_nograd_fused(case, mode):
    """Eval-mode, no-grad forward: BatchNorm (running statistics) + residual + ReLU ride in the convolution epilogue
    (vqseg_conv2d_affine_f).  Same reference and tolerances as the unfused forward."""
    from vq_seg_amd import nnf
    import copy
    n, cin, cout, h, w, k, s, p, reflect, c2, use_res = case
    if mode == "fast" and (cin % 8 or c2 % 8 or cout % 8):
        pytest.skip("bf16 mode needs channel counts that are multiples of 8")
    seed = sum(case[:8]) + 11
    conv = nn.Conv2d(cin + c2, cout, k, s, p, bias=False, padding_mode="reflect" if reflect else "zeros")
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(synth.uniform(seed, tuple(conv.weight.shape), -1, 1) * (2.0 / ((cin + c2) * k * k)) ** 0.5)
        bn.weight.copy_(synth.uniform(seed + 1, (cout,), 0.5, 1.5))
        bn.bias.copy_(synth.uniform(seed + 2, (cout,), -0.3, 0.3))
        bn.running_mean.copy_(synth.uniform(seed + 3, (cout,), -0.2, 0.2))
        bn.running_var.copy_(synth.uniform(seed + 4, (cout,), 0.5, 1.5))
    conv.eval(), bn.eval()
    dt = torch.float32 if mode == "precise" else torch.bfloat16
    x = synth.uniform(seed + 5, (n, cin, h, w), -1, 1).to(dt).float()
    x2 = synth.uniform(seed + 6, (n, c2, h, w), -1, 1).to(dt).float() if c2 else None
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    res = synth.uniform(seed + 7, (n, cout, ho, wo), -1, 1).to(dt).float() if use_res else None
    with torch.no_grad():
        ref = F.relu(bn(conv(torch.cat((x, x2), 1) if c2 else x)) + (res if use_res else 0))
        conv_g, bn_g = copy.deepcopy(conv).to(dev()), copy.deepcopy(bn).to(dev())
        out = nnf.conv_bn_act(cl(x).to(dt), conv_g, bn_g, relu=True, residual=cl(res).to(dt) if use_res else None,
                              x2=cl(x2).to(dt) if c2 else None)
        lin = nnf.conv_bn_act(cl(x).to(dt), conv_g, bn_g, relu=False, x2=cl(x2).to(dt) if c2 else None)
        ref_lin = bn(conv(torch.cat((x, x2), 1) if c2 else x))
    assert out.dtype == dt and not out.requires_grad
    tol = 2e-4 if mode == "precise" else 3e-2
    assert rel(out.float(), ref) < tol
    assert rel(lin.float(), ref_lin) < tol
    assert torch.equal(bn_g.running_mean.cpu(), bn.running_mean) and int(bn_g.num_batches_tracked) == 0


@pytest.fixture
def force_patch_kernel():
    """Let the patch-reuse 3x3 kernel take small grids too (its dispatch normally wants >= 256 workgroups)."""
    from vq_seg_amd import _hip
    L = _hip.lib()
    prev = L.vqseg_set_option(b"conv3x3_patch_min_workgroups", 1)
    assert prev >= 0, L.vqseg_last_error()
    yield
    L.vqseg_set_option(b"conv3x3_patch_min_workgroups", prev)


PATCH_CASES = [
    # n, c1, c2, cout, h, w, reflect
    (2, 64, 0, 128, 8, 32, False), (1, 128, 64, 256, 16, 64, True), (3, 64, 0, 128, 16, 16, True), (2, 192, 0, 128, 32, 48, False),
    (1, 64, 64, 384, 16, 32, False), (2, 128, 0, 256, 16, 32, False),
    (2, 64, 0, 64, 8, 32, True), (1, 128, 64, 32, 16, 32, False), (2, 64, 0, 32, 16, 16, True), (2, 128, 0, 64, 32, 32, False),
    # 32-channel chunks (channel counts that are not multiples of 64)
    (2, 32, 0, 32, 16, 32, False), (1, 96, 0, 128, 16, 16, True), (2, 32, 0, 64, 8, 32, True), (1, 32, 32, 128, 16, 32, False),
    (2, 160, 0, 32, 8, 32, False),
    # chunk stages (all nine taps' weights with the patch): single chunk x 128 / 64 / 32 outputs, several chunks x 64 / 32
    (2, 32, 0, 128, 16, 32, False), (1, 32, 0, 256, 8, 32, True), (1, 96, 0, 64, 16, 16, False), (2, 32, 64, 32, 16, 32, True),
]


TALL_CASES = [
    # 512-pixel tiles (16 x 32 or 32 x 16 pixels; 32- / 64-channel outputs, 32-channel chunks): n, c1, c2, cout, h, w, reflect
    (1, 128, 64, 32, 16, 32, False), (2, 128, 0, 64, 32, 32, False), (2, 32, 0, 32, 16, 32, False), (2, 32, 64, 32, 16, 32, True),
    (1, 64, 0, 64, 32, 16, True), (2, 192, 0, 32, 32, 64, False), (1, 64, 64, 64, 16, 64, True), (3, 96, 0, 32, 32, 48, True),
]


@pytest.mark.parametrize("case", TALL_CASES)
def test_conv3x3_patch_kernel_512_pixel_tiles(case, force_patch_kernel):
    """The same check on the 512-pixel-tile instantiations (default for 32 / 64 output channels once the grid is large enough; here
    the grid bar is lowered), with the dispatch confirmed by the library's launch counter; BatchNorm partials keep their slot size
    (64 rows from 64 channels on, else 32)."""
    from vq_seg_amd import _hip
    L = _hip.lib()
    prev = L.vqseg_set_option(b"conv3x3_patch_tile512_min_workgroups", 1)
    L.vqseg_set_option(b"conv3x3_patch_tile512_launches", 0)
    try:
        _patch_case(case)
    finally:
        L.vqseg_set_option(b"conv3x3_patch_tile512_min_workgroups", prev)
    assert L.vqseg_set_option(b"conv3x3_patch_tile512_launches", 0) >= 2, "the 512-pixel tile was not dispatched"


@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv3x3_patch_kernel(case, force_patch_kernel):
    _patch_case(case)


def _patch_case(case):
    """The patch-reuse bf16 3x3 kernel through the C ABI (forward form and, with tap-flipped transposed weights, the
    data-gradient form) against an fp64 convolution of the SAME bf16 values: products are exact in fp32, only the
    accumulation order differs, so the bf16-rounded outputs agree to one bf16 ulp (2^-8 relative)."""
    from vq_seg_amd import _hip
    n, c1, c2, cout, h, w, reflect = case
    cin = c1 + c2
    L = _hip.lib()
    seed = sum(case[:6]) + 3
    x = synth.uniform(seed, (n, h, w, cin), -1, 1).bfloat16()
    wt = (synth.uniform(seed + 1, (cout, cin, 3, 3), -1, 1) * (2.0 / (cin * 9)) ** 0.5).bfloat16().float()
    xp = F.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = F.conv2d(xp, wt.double()).permute(0, 2, 3, 1)
    wd = wt.to(dev())
    ne = L.vqseg_conv_packed_elems(cout, cin, 3, 3, 0)
    hi = torch.empty(ne, dtype=torch.int16, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    assert L.vqseg_conv_pack_weights_f32(wd.data_ptr(), cout, cin, 3, 3, 0, hi.data_ptr(), None, st) == 0
    xa = x[..., :c1].contiguous().to(dev())
    xb = x[..., c1:].contiguous().to(dev()) if c2 else None
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device=dev())
    slots = L.vqseg_conv_stat_slots(n * h * w, cout)
    stat = torch.full((slots, 2, cout), float("nan"), dtype=torch.float32, device=dev())
    rc = L.vqseg_conv2d_f(xa.data_ptr(), xb.data_ptr() if c2 else None, c1, hi.data_ptr(), None, y.data_ptr(), stat.data_ptr(),
                          n, h, w, cin, cout, 3, 3, 1, 1, int(reflect), 1, h, w, 0, st)
    assert rc == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    assert rel(y.float(), ref) < 2 ** -7
    # 1-D launch with the Cout chunks of a pixel tile on one XCD (conv3x3_patch_xcd_pair): same workgroups, same bits
    y2 = torch.full_like(y, float("nan"))
    stat2 = torch.full_like(stat, float("nan"))
    prev = L.vqseg_set_option(b"conv3x3_patch_xcd_pair", 1)
    rc = L.vqseg_conv2d_f(xa.data_ptr(), xb.data_ptr() if c2 else None, c1, hi.data_ptr(), None, y2.data_ptr(), stat2.data_ptr(),
                          n, h, w, cin, cout, 3, 3, 1, 1, int(reflect), 1, h, w, 0, st)
    L.vqseg_set_option(b"conv3x3_patch_xcd_pair", prev)
    assert rc == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y2, y) and torch.equal(stat2[:n * h * w // (64 if cout >= 64 else 32)], stat[:n * h * w // (64 if cout >= 64 else 32)])
    # BatchNorm partials of the fp32 accumulators: merged mean / biased variance per channel
    m = n * h * w
    rps = 64 if cout >= 64 else 32                         # rows per statistics slot (vqseg_conv_stat_slots)
    n_slots = m // rps
    sp = stat[:n_slots].double().cpu()
    mean = sp[:, 0].mean(0)
    var = (sp[:, 1] + rps * (sp[:, 0] - mean) ** 2).sum(0) / m
    flat = ref.reshape(m, cout)
    assert ((mean - flat.mean(0)).abs().max() / flat.abs().max()).item() < 1e-5
    assert ((var - flat.var(0, unbiased=False)).abs().max() / flat.var(0, unbiased=False).max()).item() < 1e-4
    # data-gradient form: conv of gy with the tap-flipped, transposed weights == conv_transpose (zero padding only)
    if not reflect and c2 == 0:
        gy = synth.uniform(seed + 2, (n, h, w, cout), -1, 1).bfloat16()
        gref = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), wt.double(), padding=1).permute(0, 2, 3, 1)
        ne = L.vqseg_conv_packed_elems(cout, cin, 3, 3, 1)
        thi = torch.empty(ne, dtype=torch.int16, device=dev())
        assert L.vqseg_conv_pack_weights_f32(wd.data_ptr(), cout, cin, 3, 3, 1, thi.data_ptr(), None, st) == 0
        gx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev())
        gyd = gy.to(dev())
        rc = L.vqseg_conv2d_f(gyd.data_ptr(), None, cout, thi.data_ptr(), None, gx.data_ptr(), None, n, h, w, cout, cin, 3, 3, 1, 1, 0,
                              1, h, w, 0, st)
        assert rc == 0, L.vqseg_last_error()
        torch.cuda.synchronize()
        assert rel(gx.float(), gref) < 2 ** -7


WGRAD_CASES = [
    # n, c1, c2, cout, h, w, reflect
    (2, 64, 0, 128, 8, 32, True), (3, 128, 64, 64, 12, 16, False), (2, 32, 0, 32, 16, 48, False), (1, 96, 32, 32, 8, 16, False),
    (2, 64, 0, 32, 20, 16, True), (3, 32, 0, 128, 4, 16, False), (2, 32, 0, 64, 8, 16, True), (5, 256, 256, 128, 16, 16, False),
    (4, 128, 0, 128, 32, 32, False), (4, 64, 64, 256, 32, 32, True),      # r4: >= 8 slabs of several (ci, co) tiles: the XCD-aware 1-D grid
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad3x3_fused_taps(case):
    """The fused nine-tap bf16 weight-gradient kernel through the C ABI against an fp64 reference on the SAME bf16
    values: products of bf16 pairs are exact in fp32, so only the accumulation order differs (tolerance 2e-5 of
    the tensor scale).  Reference: torch.nn.grad.conv2d_weight on the explicitly padded input."""
    from vq_seg_amd import _hip
    n, c1, c2, cout, h, w, reflect = case
    cin = c1 + c2
    L = _hip.lib()
    seed = sum(case[:6])
    x = synth.uniform(seed, (n, h, w, cin), -1, 1).bfloat16()
    gy = synth.uniform(seed + 1, (n, h, w, cout), -1, 1).bfloat16()
    xp = F.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = torch.nn.grad.conv2d_weight(xp, (cout, cin, 3, 3), gy.double().permute(0, 3, 1, 2))
    xa = x[..., :c1].contiguous().to(dev())
    xb = x[..., c1:].contiguous().to(dev()) if c2 else None
    gyd = gy.to(dev())
    nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(n, h, w, cin, h, w, cout, 3, 3)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    gw = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev())
    rc = L.vqseg_conv2d_wgrad_f(gyd.data_ptr(), xa.data_ptr(), xb.data_ptr() if c2 else None, c1, n, h, w, cin, h, w, cout, 3, 3,
                                1, 1, int(reflect), 0, cin, 0, 0, ws.data_ptr(), nbytes, gw.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    assert rel(gw, ref) < 2e-5
    # accumulate = 1 adds the same gradient onto gw (the trainer's bucket view)
    rc = L.vqseg_conv2d_wgrad_f(gyd.data_ptr(), xa.data_ptr(), xb.data_ptr() if c2 else None, c1, n, h, w, cin, h, w, cout, 3, 3,
                                1, 1, int(reflect), 0, cin, 0, 1, ws.data_ptr(), nbytes, gw.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.vqseg_last_error()
    assert rel(gw, 2 * ref) < 2e-5


WGRAD_S2_CASES = [
    # n, cin, cout, h, w (input size; output h / 2, w / 2), reflect, second source (two-use launch)
    (2, 128, 128, 32, 32, True, False), (3, 64, 128, 8, 64, False, False), (1, 96, 256, 12, 32, True, False),
    (2, 256, 128, 4, 32, False, True), (3, 128, 256, 16, 32, True, True),
    (4, 128, 256, 64, 64, True, False), (4, 128, 128, 64, 64, False, True),     # >= 8 slabs of several tiles: the XCD-aware 1-D grid
]


@pytest.mark.parametrize("case", WGRAD_S2_CASES)
def test_wgrad3x3_stride2_fused_taps(case):
    """Stride-2 3x3 layers (first conv2 of a Bottleneck stage, first conv1 of a BasicBlock stage) on the nine-tap kernel (r4: 2 x 16
    output blocks, the 5 x 33 input patch de-interleaved into column-parity planes by the DMA): fp64 reference on the same bf16 values,
    2e-5 of the tensor scale; and the r3 per-tap kernel (option off) agrees with the same reference."""
    from vq_seg_amd import _hip
    n, cin, cout, h, w, reflect, pair = case
    ho, wo = h // 2, w // 2
    L = _hip.lib()
    seed = sum(case[:5]) + 11
    x = synth.uniform(seed, (n, h, w, cin), -1, 1).bfloat16()
    gy = synth.uniform(seed + 1, (n, ho, wo, cout), -1, 1).bfloat16()
    xp = F.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = torch.nn.grad.conv2d_weight(xp, (cout, cin, 3, 3), gy.double().permute(0, 3, 1, 2), stride=2)
    xd, gyd = x.to(dev()), gy.to(dev())
    nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(n, h, w, cin, ho, wo, cout, 3, 3)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    for opt in (1, 0):
        prev = _hip.set_option("conv_wgrad3x3_stride2", opt)
        try:
            gw = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev())
            if pair:                                    # images [0, na) from the first source, the rest from the second
                na = 1
                rc = L.vqseg_conv2d_wgrad2_f(gyd[:na].data_ptr(), xd[:na].data_ptr(), None, na, gyd[na:].data_ptr(), xd[na:].data_ptr(), None,
                                             n - na, cin, h, w, cin, ho, wo, cout, 3, 3, 2, 1, int(reflect), 0, cin, 0, 0,
                                             ws.data_ptr(), nbytes, gw.data_ptr(), st)
            else:
                rc = L.vqseg_conv2d_wgrad_f(gyd.data_ptr(), xd.data_ptr(), None, cin, n, h, w, cin, ho, wo, cout, 3, 3, 2, 1, int(reflect),
                                            0, cin, 0, 0, ws.data_ptr(), nbytes, gw.data_ptr(), st)
            assert rc == 0, L.vqseg_last_error()
            torch.cuda.synchronize()
            assert rel(gw, ref) < 2e-5, opt
        finally:
            _hip.set_option("conv_wgrad3x3_stride2", prev)


WGRAD1_CASES = [
    # n, cin, cout, h, w, stride   (every tile config of the 1x1 weight-gradient kernel; ragged pixel counts; stride 2)
    (2, 128, 256, 16, 16, 1), (3, 64, 256, 9, 7, 1), (2, 128, 128, 8, 24, 1), (1, 64, 128, 20, 20, 1), (2, 256, 64, 12, 12, 1),
    (2, 256, 512, 16, 16, 2), (1, 128, 256, 15, 9, 2), (4, 1024, 256, 8, 8, 1),
    (2, 64, 64, 16, 16, 1), (3, 64, 64, 9, 7, 1), (2, 192, 64, 12, 20, 1),      # r4: 64 output channels on four waves (<2, 2>)
    (2, 256, 256, 32, 32, 1), (2, 256, 512, 64, 32, 2),                         # r4: >= 8 slabs of several tiles: the XCD-aware 1-D grid
]


@pytest.mark.parametrize("case", WGRAD1_CASES)
def test_wgrad1x1(case):
    """bf16 1x1 weight-gradient kernel through the C ABI against fp64 on the same bf16 values (2e-5 of scale)."""
    from vq_seg_amd import _hip
    n, cin, cout, h, w, stride = case
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    L = _hip.lib()
    seed = sum(case) + 5
    x = synth.uniform(seed, (n, h, w, cin), -1, 1).bfloat16()
    gy = synth.uniform(seed + 1, (n, ho, wo, cout), -1, 1).bfloat16()
    xs = x[:, ::stride, ::stride, :].double().reshape(-1, cin)
    ref = (gy.double().reshape(-1, cout).t() @ xs).reshape(cout, cin, 1, 1)
    xd, gyd = x.to(dev()), gy.to(dev())
    nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(n, h, w, cin, ho, wo, cout, 1, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    gw = torch.full((cout, cin, 1, 1), float("nan"), dtype=torch.float32, device=dev())
    for acc in (0, 1):
        rc = L.vqseg_conv2d_wgrad_f(gyd.data_ptr(), xd.data_ptr(), None, cin, n, h, w, cin, ho, wo, cout, 1, 1, stride, 0, 0, 0, cin, 0,
                                    acc, ws.data_ptr(), nbytes, gw.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, L.vqseg_last_error()
        torch.cuda.synchronize()
        assert rel(gw, (1 + acc) * ref) < 2e-5


@pytest.mark.parametrize("reflect", [True, False])
@pytest.mark.parametrize("shape", [(2, 40, 36), (1, 8, 300), (3, 130, 128)])
def test_stem_patch_matrix_strip_kernel_is_bit_identical_to_the_gather_kernel(shape, reflect):
    """r4: the stem's im2col patch matrix from LDS-staged strips of 64 output pixels (7 input rows per strip, padding applied while
    staging) -- every output type (fp32 / bf16 / split-3 rows), ragged last strips, both paddings: the same bits as the r3 gather
    kernel (option off), and the fp32 form equals F.unfold on the padded image (reference: resnet.py:122-125's 7x7 / 2 / 3 conv)."""
    from vq_seg_amd import _hip
    n, h, w = shape
    ho, wo, kp = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1, 160
    L = _hip.lib()
    x = synth.uniform(n + h + w, (n, h, w, 3), -1, 1)
    xd = x.to(dev())
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for opt in (1, 0):
        prev = _hip.set_option("im2col_strip", opt)
        try:
            for kind, dt, width, kpk in ((0, torch.float32, kp, kp), (1, torch.bfloat16, kp, kp), (2, torch.bfloat16, 2 * kp, kp),
                                         (3, torch.bfloat16, 2 * 192, 192)):        # 3: split-3 rows of 192 columns (the s3 forward's width)
                out = torch.full((n * ho * wo, width), float("nan"), dtype=dt, device=dev())
                rc = L.vqseg_im2col_f(min(kind, 2), xd.data_ptr(), n, h, w, 3, 7, 7, 2, 3, int(reflect), ho, wo, kpk, out.data_ptr(), st)
                assert rc == 0, L.vqseg_last_error()
                torch.cuda.synchronize()
                outs[(opt, kind)] = out.cpu()
        finally:
            _hip.set_option("im2col_strip", prev)
    for kind in (0, 1, 2, 3):
        assert torch.equal(outs[(1, kind)].view(torch.int16 if kind else torch.int32), outs[(0, kind)].view(torch.int16 if kind else torch.int32)), kind
    xp = F.pad(x.permute(0, 3, 1, 2), (3, 3, 3, 3), mode="reflect" if reflect else "constant")
    ref = F.unfold(xp, 7, stride=2).reshape(n, 3, 7, 7, ho * wo).permute(0, 4, 2, 3, 1).reshape(n * ho * wo, 147)   # columns (kh, kw, ci)
    assert torch.equal(outs[(1, 0)][:, :147], ref) and not outs[(1, 0)][:, 147:].any()


@pytest.mark.parametrize("pair", [False, True])
def test_wgrad_stem_patch_matrix(pair):
    """The stem's weight gradient = a 1x1 weight gradient over its im2col patch matrix ([rows][160 columns = (kh, kw, ci) padded]).
    r4: all 160 columns in ONE ten-wave workgroup of the LDS-DMA kernel (<2, 5>, 320-byte rows) instead of five 32-column tiles of the
    per-tap kernel; fp64 reference on the same bf16 values, 2e-5 of the scale; the per-tap kernel (option off) meets the same bar."""
    from vq_seg_amd import _hip
    n, ho, wo, cout, kp = 3, 10, 13, 64, 160                # ragged row count: 390 rows = 6 stages + 6 rows
    L = _hip.lib()
    pm = synth.uniform(31, (n, ho, wo, kp), -1, 1).bfloat16()
    pm[..., 147:] = 0                                       # the padding columns of a real patch matrix
    gy = synth.uniform(32, (n, ho, wo, cout), -1, 1).bfloat16()
    full = gy.double().reshape(-1, cout).t() @ pm.double().reshape(-1, kp)           # [co][(kh, kw, ci)]
    ref = full[:, :147].reshape(cout, 7, 7, 3).permute(0, 3, 1, 2).contiguous()
    pmd, gyd = pm.to(dev()), gy.to(dev())
    nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(n, ho, wo, kp, ho, wo, cout, 1, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    for opt in (1, 0):
        prev = _hip.set_option("conv_wgrad1x1_narrow", opt)
        try:
            gw = torch.full((cout, 3, 7, 7), float("nan"), dtype=torch.float32, device=dev())
            for acc in (0, 1):
                if pair:
                    rc = L.vqseg_conv2d_wgrad2_f(gyd[:2].data_ptr(), pmd[:2].data_ptr(), None, 2, gyd[2:].data_ptr(), pmd[2:].data_ptr(), None, 1,
                                                 kp, ho, wo, kp, ho, wo, cout, 7, 7, 1, 0, 0, 0, 3, 1, acc, ws.data_ptr(), nbytes, gw.data_ptr(), st)
                else:
                    rc = L.vqseg_conv2d_wgrad_f(gyd.data_ptr(), pmd.data_ptr(), None, kp, n, ho, wo, kp, ho, wo, cout, 7, 7, 1, 0, 0, 0, 3, 1,
                                                acc, ws.data_ptr(), nbytes, gw.data_ptr(), st)
                assert rc == 0, L.vqseg_last_error()
                torch.cuda.synchronize()
                assert rel(gw, (1 + acc) * ref) < 2e-5, (opt, acc)
        finally:
            _hip.set_option("conv_wgrad1x1_narrow", prev)


@pytest.mark.parametrize("mode", ["precise", "fast"])
@pytest.mark.parametrize("reflect", [True, False])
def test_stem(reflect, mode):
    from vq_seg_amd import nnf
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False, padding_mode="reflect" if reflect else "zeros")
    bn = nn.BatchNorm2d(64)
    x = synth.uniform(5, (2, 3, 40, 36))
    g = synth.uniform(6, (2, 64, 20, 18), -1, 1)
    if mode == "fast":
        g = g.bfloat16().float()
    with torch.no_grad():                                          # see test_conv_bn_act: zero the cotangent near the ReLU kink
        pre = bn(conv(x))
        bn.running_mean.zero_(), bn.running_var.fill_(1.0), bn.num_batches_tracked.zero_()
    g = g * (pre.abs() > (1e-4 if mode == "precise" else 5e-2)).float()
    y = F.relu(bn(conv(x)))
    y.backward(g)
    import copy
    cg, bg = copy.deepcopy(conv).to(dev()), copy.deepcopy(bn).to(dev())
    cg.weight.grad = None
    bg.weight.grad = bg.bias.grad = None
    with torch.no_grad():
        bg.running_mean.zero_(), bg.running_var.fill_(1.0)
    if mode == "fast":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = nnf.stem_conv_bn_act(cl(x), cg, bg)
    else:
        out = nnf.stem_conv_bn_act(cl(x), cg, bg)
    out.backward(cl(g).to(out.dtype))
    tol = 2e-4 if mode == "precise" else 3e-2
    assert out.dtype == (torch.float32 if mode == "precise" else torch.bfloat16)
    assert rel(out.float(), y) < tol
    assert rel(cg.weight.grad, conv.weight.grad) < tol
    assert rel(bg.weight.grad, bn.weight.grad) < tol and rel(bg.bias.grad, bn.bias.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_bilinear_head(dtype):
    from vq_seg_amd import nnf
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    x = synth.uniform(1, (2, 64, 18, 22), -1, 1)
    xq = x.to(dtype).float()                                   # the values the kernel actually sees
    g = synth.uniform(2, (2, 64, 9, 11), -1, 1)
    xr = xq.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    y.backward(g.to(dtype).float())
    xg = cl(x).to(dtype).requires_grad_(True)
    out = nnf.max_pool_3x3_s2(xg)
    out.backward(cl(g).to(dtype))
    assert rel(out.float(), y) < tol and rel(xg.grad.float(), xr.grad) < tol
    for align, (ho, wo) in ((False, (36, 44)), (True, (36, 44)), (False, (35, 41))):
        xr = xq.clone().requires_grad_(True)
        y = F.interpolate(xr, size=(ho, wo), mode="bilinear", align_corners=align)
        gg = synth.uniform(3, tuple(y.shape), -1, 1)
        y.backward(gg.to(dtype).float())
        xg = cl(x).to(dtype).requires_grad_(True)
        out = nnf.upsample_bilinear(xg, size=(ho, wo), align_corners=align)
        out.backward(cl(gg).to(dtype))
        assert rel(out.float(), y) < max(tol, 1e-6), (align, ho, wo)
        assert rel(xg.grad.float(), xr.grad) < max(tol, 2e-6), (align, ho, wo)
    # the logits' shapes: 3 channels (one pixel per thread) and another odd count (one element per thread), fp32, x2 upsampling
    if dtype == torch.float32:
        for ch in (3, 5):
            x3 = synth.uniform(7, (2, ch, 20, 24), -1, 1)
            xr = x3.clone().requires_grad_(True)
            y = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
            gg = synth.uniform(8, tuple(y.shape), -1, 1)
            y.backward(gg)
            xg = cl(x3).requires_grad_(True)
            out = nnf.upsample_bilinear(xg, scale_factor=2, align_corners=True)
            out.backward(cl(gg))
            assert rel(out, y) < 1e-6 and rel(xg.grad, xr.grad) < 2e-6, ch
    # 1x1 head: fp32 logits
    w = synth.uniform(4, (3, 32, 1, 1), -1, 1)
    x = synth.uniform(5, (2, 32, 20, 20), -1, 1)
    xq = x.to(dtype).float()
    xr, wr = xq.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr)
    gg = synth.uniform(6, tuple(y.shape), -1, 1)
    y.backward(gg)
    xg, wg = cl(x).to(dtype).requires_grad_(True), w.to(dev()).requires_grad_(True)
    out = nnf.head_conv1x1(xg, wg)
    assert out.dtype == torch.float32
    out.backward(cl(gg))
    assert rel(out, y) < 1e-5 and rel(wg.grad, wr.grad) < 1e-5 and rel(xg.grad.float(), xr.grad) < max(tol, 1e-5)


def test_kernels_are_deterministic():
    from vq_seg_amd import nnf
    conv = nn.Conv2d(64, 128, 3, 1, 1, bias=False).to(dev())
    bn = nn.BatchNorm2d(128).to(dev())
    x = cl(synth.uniform(1, (4, 64, 32, 32), -1, 1))
    outs = []
    for _ in range(3):
        xg = x.clone().requires_grad_(True)
        conv.weight.grad = None
        o = nnf.conv_bn_act(xg, conv, bn)
        o.sum().backward()
        outs.append((o.detach().clone(), xg.grad.clone(), conv.weight.grad.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0])), "run-to-run differences"


def test_grad_sink_accumulates_in_place():
    """A parameter marked with `_vq_grad_sink` receives its gradient by in-place accumulation from the HIP kernels:
    two uses of the same conv+bn in one graph add up in p.grad exactly like autograd's own accumulation, the callback
    fires once per parameter (after the last contribution), and autograd adds nothing on top."""
    from vq_seg_amd import nnf
    import copy
    torch.manual_seed(0)
    conv = nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(dev())
    bn = nn.BatchNorm2d(64).to(dev())
    conv2, bn2 = copy.deepcopy(conv), copy.deepcopy(bn)
    xa = cl(synth.uniform(1, (2, 64, 16, 16), -1, 1))
    xb = cl(synth.uniform(2, (2, 64, 16, 16), -1, 1))
    # plain autograd accumulation
    (nnf.conv_bn_act(xa, conv, bn).square().sum() + nnf.conv_bn_act(xb, conv, bn).sum()).backward()
    # sinks
    fired = []
    for p_ in list(conv2.parameters()) + list(bn2.parameters()):
        p_.grad = torch.full_like(p_, 0.5)
        p_._vq_grad_sink = lambda q: fired.append(id(q))
    (nnf.conv_bn_act(xa, conv2, bn2).square().sum() + nnf.conv_bn_act(xb, conv2, bn2).sum()).backward()
    assert sorted(fired) == sorted(id(q) for q in list(conv2.parameters()) + list(bn2.parameters()))
    for a_, b_ in zip(list(conv.parameters()) + list(bn.parameters()), list(conv2.parameters()) + list(bn2.parameters())):
        assert rel(b_.grad - 0.5, a_.grad) < 1e-5


@pytest.mark.parametrize("shortcut", ["identity", "projection", "projection_stride2"])
@pytest.mark.parametrize("mode", ["precise", "fast"])
def test_bottleneck_grad_link_is_exact(mode, shortcut):
    """Bottleneck: folding the shortcut branch's gradient of the block input into conv1's data-gradient epilogue
    (nnf.GradLink; identity shortcut: the block's output gradient, projection shortcut: the projection's data gradient)
    must give bit-identical gradients to autograd's separate add (same roundings in the same order)."""
    from torch import nn
    from vq_seg_amd import nnf
    from vq_seg_amd.models.encoders.resnet import Bottleneck
    torch.manual_seed(1)
    if shortcut == "identity":
        blk, cin, so = Bottleneck(256, 64), 256, 16
    else:
        stride = 2 if shortcut.endswith("2") else 1
        blk = Bottleneck(128, 64, stride, nn.Sequential(nn.Conv2d(128, 256, 1, stride, bias=False), nn.BatchNorm2d(256)))
        cin, so = 128, 16 // stride
    blk = blk.to(dev())
    blk.train()
    dt = torch.float32 if mode == "precise" else torch.bfloat16
    x0 = cl(synth.uniform(3, (2, cin, 16, 16), -1, 1)).to(dt)
    g = cl(synth.uniform(4, (2, 256, so, so), -1, 1)).to(dt)

    def run(use_link):
        saved = nnf.GradLink
        if not use_link:
            nnf.GradLink = lambda: None
        try:
            for p_ in blk.parameters():
                p_.grad = None
            x = x0.clone().requires_grad_(True)
            blk(x).backward(g)
            return [x.grad.clone()] + [p_.grad.clone() for p_ in blk.parameters()]
        finally:
            nnf.GradLink = saved

    a, b = run(True), run(False)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    if shortcut != "identity":                                    # the fused path was really taken: no producer came late
        x = x0.clone().requires_grad_(True)
        seen = []
        saved = nnf.GradLink

        class Spy(saved):
            def __init__(self):
                self.sets = 0
                super().__init__()
                seen.append(self)

            @property
            def g(self):
                return self._g

            @g.setter
            def g(self, v):
                self._g = v
                self.sets += v is not None
        nnf.GradLink = Spy
        try:
            blk(x).backward(g)
        finally:
            nnf.GradLink = saved
        assert len(seen) == 1 and seen[0].closed and seen[0].g is None and seen[0].sets == 1


def test_short_k_dispatch_variants_agree():
    """1x1 layers with a short K loop may take the single-buffered 128x128 tile (conv_short_k_single_buffer) or the 4-waves/SIMD
    double-buffered one (conv_short_k_small_tile): same accumulation order, so the outputs must be bit-identical to the default."""
    from vq_seg_amd import _hip
    L = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    n, h, w = 2, 24, 20
    for cin, cout in ((64, 256), (128, 512), (256, 128), (512, 256)):
        x = synth.uniform(cin + cout, (n, h, w, cin), -1, 1).bfloat16().to(dev())
        wt = (synth.uniform(cin, (cout, cin, 1, 1), -1, 1) * (2.0 / cin) ** 0.5).to(dev())
        hi = torch.empty(L.vqseg_conv_packed_elems(cout, cin, 1, 1, 0), dtype=torch.int16, device=dev())
        assert L.vqseg_conv_pack_weights_f32(wt.data_ptr(), cout, cin, 1, 1, 0, hi.data_ptr(), None, st) == 0
        outs = []
        for opts in ({}, {"conv_short_k_single_buffer": 8, "conv_short_k_small_tile": 0}, {"conv_short_k_small_tile": 8},
                     {"conv_xcd_pair": 0}, {"conv_xcd_pair": 16}):       # 2-D grid / Cout chunks of an M tile on one XCD
            prev = {k: L.vqseg_set_option(k.encode(), v) for k, v in opts.items()}
            y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device=dev())
            rc = L.vqseg_conv2d_f(x.data_ptr(), None, cin, hi.data_ptr(), None, y.data_ptr(), None, n, h, w, cin, cout, 1, 1, 1, 0, 0, 1, h, w, 0, st)
            for k, v in prev.items():
                L.vqseg_set_option(k.encode(), v)
            assert rc == 0, L.vqseg_last_error()
            outs.append(y)
        torch.cuda.synchronize()
        ref = (x.float().reshape(-1, cin) @ wt.reshape(cout, cin).bfloat16().float().t()).reshape(n, h, w, cout)
        assert rel(outs[0].float(), ref) < 2 ** -7
        assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_stem_patch_matrix_is_shared_inside_a_scope_only():
    """nnf.stem_share_begin/end: stems that unfold the SAME image tensor (same object, same version) inside the scope reuse one
    patch matrix -- across modules and streams; a different tensor, an in-place change of the tensor, or the end of the scope
    each get a fresh unfold.  Outputs are bit-identical to the unshared path."""
    from vq_seg_amd import nnf
    torch.manual_seed(0)
    conv_a, bn_a = nn.Conv2d(3, 64, 7, 2, 3, bias=False, padding_mode="reflect").to(dev()), nn.BatchNorm2d(64).to(dev())
    conv_b, bn_b = nn.Conv2d(3, 64, 7, 2, 3, bias=False, padding_mode="reflect").to(dev()), nn.BatchNorm2d(64).to(dev())
    x = cl(synth.uniform(1, (2, 3, 64, 64), 0, 1))
    y = cl(synth.uniform(2, (2, 3, 64, 64), 0, 1))
    with torch.no_grad():
        ref_a, ref_b = nnf.stem_conv_bn_act(x, conv_a, bn_a), nnf.stem_conv_bn_act(x, conv_b, bn_b)
        assert nnf._STEM_SHARE is None
        nnf.stem_share_begin()
        try:
            out_a = nnf.stem_conv_bn_act(x, conv_a, bn_a)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out_b = nnf.stem_conv_bn_act(x, conv_b, bn_b)               # other module, other stream: same matrix
            torch.cuda.current_stream().wait_stream(side)
            assert len(nnf._STEM_SHARE) == 1
            nnf.stem_conv_bn_act(y, conv_a, bn_a)
            assert len(nnf._STEM_SHARE) == 2
            x.mul_(0.5)                                                     # same object, new version: unfolded again
            out_c = nnf.stem_conv_bn_act(x, conv_a, bn_a)
            assert len(nnf._STEM_SHARE) == 3
        finally:
            nnf.stem_share_end()
        assert nnf._STEM_SHARE is None
        ref_c = nnf.stem_conv_bn_act(x, conv_a, bn_a)
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b) and torch.equal(out_c, ref_c)


def test_wrong_element_types_raise_in_python_instead_of_faulting_on_the_gpu():
    """the C ABI takes untyped pointers + a bf16 flag; every tensor is handed over through _hip.tptr,
    which checks the element type against that flag and the element count against the sizes -- on the GPU box a mix-up is an exception
    (the r2 fault was a bf16 buffer read as f32: 2x out of bounds)."""
    import torch
    from torch import nn
    from vq_seg_amd import _hip, nnf
    d = torch.device("cuda:0")
    E = _hip.HipLibraryError
    xb = torch.zeros(2, 32, 8, 8, device=d, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xf = torch.zeros(2, 32, 8, 8, device=d).contiguous(memory_format=torch.channels_last)
    conv, bn = nn.Conv2d(32, 32, 3, padding=1, bias=False).to(d), nn.BatchNorm2d(32).to(d)
    with pytest.raises(E, match="residual dtype differs|expected torch.bfloat16"):
        nnf.conv_bn_act(xb, conv, bn, residual=xf)
    with pytest.raises(E, match="conv input 2: expected torch.bfloat16"):
        conv2 = nn.Conv2d(64, 32, 3, padding=1, bias=False).to(d)
        nnf.conv_bn_act(xb, conv2, bn, x2=xf)
    with pytest.raises(E, match="rows: expected torch.float32"):
        _hip.vq_assign(torch.zeros(64, 32, device=d, dtype=torch.float16), torch.zeros(8, 32, device=d))
    with pytest.raises(E, match="codebook: the sizes passed along need"):
        _hip.vq_assign(torch.zeros(64, 32, device=d), torch.zeros(8, 64, device=d))
    with pytest.raises(E, match="need 4096 elements"):
        nnf.S3(torch.zeros(2, 8, 8, 16, device=d, dtype=torch.bfloat16), 16).float()
    bn.double()
    with pytest.raises(E, match="bn.weight: expected torch.float32"):
        nnf.conv_bn_act(xf, conv, bn)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_one_launch_batchnorm_forms_match_the_two_launch_forms(dtype):
    """The opt-in one-launch statistics merge + finalize / backward reduce + finalize (include/vqseg.h `sync`: last-workgroup
    hand-over through device-coherent accesses; VQSEG_OPTS=py_bn_fused=1) against the default two-launch forms on a tall tensor
    (enough conv-epilogue slots for the two-level merge, several row blocks per channel group): forward statistics, running
    statistics and outputs bit-identical (same fold order); parameter / input gradients to 1e-5 (the row-block partition differs);
    twice in a row (the hand-over counters must come back to zero)."""
    from vq_seg_amd import _hip, nnf
    torch.manual_seed(4)
    d = dev()
    x = synth.relu_features(31, (8, 64, 96, 96)).to(d).to(dtype).contiguous(memory_format=torch.channels_last)
    g = synth.uniform(32, (8, 128, 96, 96), -1, 1).to(d).to(dtype).contiguous(memory_format=torch.channels_last)

    def run(fused):
        conv = nn.Conv2d(64, 128, 3, padding=1, bias=False).to(d)
        bn = nn.BatchNorm2d(128).to(d)
        with torch.no_grad():
            conv.weight.copy_(synth.uniform(33, (128, 64, 3, 3), -0.05, 0.05).to(d))
            bn.weight.copy_(synth.uniform(34, (128,), 0.5, 1.5).to(d))
        _hip.lib()
        prev = _hip.PY_OPTS.get("py_bn_fused")
        _hip.PY_OPTS["py_bn_fused"] = int(fused)
        outs = []
        try:
            for _ in range(2):
                xx = x.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                    y = nnf.conv_bn_act(xx, conv, bn, relu=True)
                y.backward(g)
                outs.append((y.detach().clone(), xx.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                             bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()))
                conv.weight.grad = bn.weight.grad = bn.bias.grad = None
        finally:
            if prev is None:
                _hip.PY_OPTS.pop("py_bn_fused")
            else:
                _hip.PY_OPTS["py_bn_fused"] = prev
        if fused:
            assert int(bn._vq_sync.abs().sum()) == 0                        # counters back to zero
        return outs
    two, one = run(False), run(True)
    for a, b in zip(two, one):
        assert torch.equal(a[0], b[0]) and torch.equal(a[5], b[5]) and torch.equal(a[6], b[6]) and torch.equal(a[7], b[7])
        for i in (1, 2, 3, 4):
            assert rel(b[i].float(), a[i].float()) < 1e-5, i


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 8, 1, 4), (2, 128, 32, 8), (1, 1024, 2, 2), (1, 16, 4, 8), (2, 32, 8, 2)])
def test_exact_2x_bilinear_kernels_are_bit_identical_to_the_generic_ones(shape, dtype):
    """The decoder's resize to a skip of twice the size (align_corners = False) takes bilinear_up2_{fwd,bwd}_kernel on power-of-two
    extents: same expressions, values and order as the generic kernels (`vqseg_set_option("bilinear_up2", 0)`), so forward and
    backward must agree bit for bit -- borders (clamped taps folding onto one row / column, extents of 1 and 2) included -- and with
    ATen's upsample_bilinear2d to the usual tolerance."""
    from vq_seg_amd import _hip, nnf
    L = _hip.lib()
    n, c, h, w = shape
    x = synth.uniform(41, shape, -2, 2).to(dev()).to(dtype).contiguous(memory_format=torch.channels_last)
    g = synth.uniform(42, (n, c, 2 * h, 2 * w), -1, 1).to(dev()).to(dtype).contiguous(memory_format=torch.channels_last)

    def run():
        xx = x.clone().requires_grad_(True)
        y = nnf.upsample_bilinear(xx, size=(2 * h, 2 * w), align_corners=False)
        y.backward(g)
        return y.detach().clone(), xx.grad.clone()
    fast = run()
    prev = L.vqseg_set_option(b"bilinear_up2", 0)
    try:
        generic = run()
        L.vqseg_set_option(b"bilinear_up2", 2)                 # r3's backward: one input row per thread (r4 default: four, extents >= 4)
        one_row = run()
    finally:
        L.vqseg_set_option(b"bilinear_up2", prev)
    assert torch.equal(fast[0], generic[0]) and torch.equal(fast[1], generic[1])
    assert torch.equal(fast[0], one_row[0]) and torch.equal(fast[1], one_row[1])
    xr = x.float().cpu().requires_grad_(True)
    yr = F.interpolate(xr, size=(2 * h, 2 * w), mode="bilinear", align_corners=False)
    yr.backward(g.float().cpu())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel(fast[0].float(), yr) < tol and rel(fast[1].float(), xr.grad) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 32, 24, 40, 3), (1, 16, 9, 7, 3), (2, 64, 16, 16, 5)])
def test_conv_with_bias_head_on_the_hip_kernels(shape):
    """nnf.conv2d_bias (r4): the plain Unet's 3x3 + bias head (segmentation_head.py:78-83; 32 -> 3 channels, fp32) on the precise-mode
    kernels, output channels padded to their 4-channel granule, bias in the fused epilogue -- forward and all three gradients against
    F.conv2d in fp64 (split bf16 hi + lo operands: 2e-5 of the scale)."""
    from vq_seg_amd import nnf
    n, cin, h, w, cout = shape
    x = synth.uniform(sum(shape), (n, cin, h, w), -1, 1)
    wt = synth.uniform(sum(shape) + 1, (cout, cin, 3, 3), -0.3, 0.3)
    b = synth.uniform(sum(shape) + 2, (cout,), -0.5, 0.5)
    g = synth.uniform(sum(shape) + 3, (n, cout, h, w), -1, 1)
    xr, wr, br = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    F.conv2d(xr, wr, br, 1, 1).backward(g.double())
    xd = cl(x).to(dev()).requires_grad_(True)
    wd, bd = wt.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    y = nnf.conv2d_bias(xd, wd, bd, 1)
    assert y.shape == (n, cout, h, w) and y.dtype == torch.float32
    y.backward(g.to(dev()))
    assert rel(y.detach(), F.conv2d(x.double(), wt.double(), b.double(), 1, 1)) < 2e-5
    assert rel(xd.grad, xr.grad) < 2e-5 and rel(wd.grad, wr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("reflect", [True, False])
def test_stem_from_the_image_matches_the_patch_matrix_path(reflect):
    """r4: the stem convolution straight from the fp32 image (vqseg_stem7_conv_f: a workgroup stages the input rows its 128 output
    pixels read; contraction over kh * 24 + (kw, ci), each kernel row padded to 24 taps) against the 1x1 convolution over the
    materialised patch matrix: the same products, another summation grouping -- training forward (output, running statistics) and
    backward (weight / BatchNorm gradients: the patch matrix is built in backward then), the eval forward with the fused epilogue, and
    the split-3 (fp32-precision) eval forward.  bf16 outputs within one unit in the last place (2^-7 of the value), 2e-3 of the
    tensor scale in rel-L2; the split-3 output (hi + lo) to 3e-5 (three bf16 products per term: ~2^-16 each); gradients and statistics to 1e-3."""
    import copy
    from vq_seg_amd import _hip, nnf
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False, padding_mode="reflect" if reflect else "zeros").to(dev())
    bn = nn.BatchNorm2d(64).to(dev())
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.3, 0.3)
    x = cl(synth.uniform(21, (3, 3, 16, 512))).to(dev())         # -> 8 x 256 output pixels: two 128-pixel strips per row
    g = cl(synth.uniform(22, (3, 64, 8, 256), -1, 1)).to(dev())
    res = {}
    for fused in (1, 0):
        prev = _hip.PY_OPTS.get("py_stem_fused")
        _hip.PY_OPTS["py_stem_fused"] = fused
        try:
            c, b = copy.deepcopy(conv), copy.deepcopy(bn)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = nnf.stem_conv_bn_act(x, c, b)
            out.backward(g.to(out.dtype))
            b.eval()
            with torch.no_grad():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    ev = nnf.stem_conv_bn_act(x, c, b)
                with nnf.s3_scope(True):
                    s3 = nnf.stem_conv_bn_act(x, c, b)
            torch.cuda.synchronize()
            s3v = s3.rows[..., :64].float() + s3.rows[..., 64:].float()
            res[fused] = (out.detach(), c.weight.grad, b.weight.grad, b.bias.grad, b.running_mean, b.running_var, ev, s3v)
        finally:
            if prev is None:
                _hip.PY_OPTS.pop("py_stem_fused", None)
            else:
                _hip.PY_OPTS["py_stem_fused"] = prev
    assert res[1][0].dtype == torch.bfloat16 and res[1][7].shape == (3, 8, 256, 64)
    tol = (2e-3, 1e-3, 1e-3, 1e-3, 1e-4, 1e-4, 2e-3, 3e-5)
    for i, (a, b_) in enumerate(zip(res[1], res[0])):
        assert a.shape == b_.shape and a.dtype == b_.dtype, i
        assert rel(a.float(), b_.float()) < tol[i], (i, rel(a.float(), b_.float()))
    for i in (0, 6):                                                # bf16 outputs: never more than one unit in the last place apart
        a, b_ = res[1][i].float(), res[0][i].float()
        assert ((a - b_).abs() <= 2 ** -7 * b_.abs() + 1e-30).all(), i


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(16, 64, 256, 64, 64, True), (16, 64, 512, 64, 64, False)])     # >= 256 (pixel tile, 256-channel chunk) workgroups: the dispatch takes the wide tile
def test_conv3x3_patch_kernel_256_channel_tile(case):
    """The 256-channel tile of the 3x3 patch kernel (default from r4 where it still fills the chip: the input rows are read once per 256
    instead of per 128 output channels) against the 128-channel tile (option off): same outputs AND BatchNorm partials, bit for bit
    (the same K order per output element); fp64 reference 2^-7 (bf16 output)."""
    from vq_seg_amd import _hip
    n, cin, cout, h, w, reflect = case
    L = _hip.lib()
    seed = sum(case[:5]) + 9
    x = synth.uniform(seed, (n, h, w, cin), -1, 1).bfloat16()
    wt = (synth.uniform(seed + 1, (cout, cin, 3, 3), -1, 1) * (2.0 / (cin * 9)) ** 0.5).bfloat16().float()
    xp = F.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = F.conv2d(xp, wt.double()).permute(0, 2, 3, 1)
    wd = wt.to(dev())
    ne = L.vqseg_conv_packed_elems(cout, cin, 3, 3, 0)
    hi = torch.empty(ne, dtype=torch.int16, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    assert L.vqseg_conv_pack_weights_f32(wd.data_ptr(), cout, cin, 3, 3, 0, hi.data_ptr(), None, st) == 0
    xa = x.to(dev())
    slots = L.vqseg_conv_stat_slots(n * h * w, cout)
    res = {}
    for wide in (1, 0):
        prev = _hip.set_option("conv3x3_patch_wide_tile", wide)
        try:
            y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device=dev())
            stat = torch.full((slots, 2, cout), float("nan"), dtype=torch.float32, device=dev())
            rc = L.vqseg_conv2d_f(xa.data_ptr(), None, cin, hi.data_ptr(), None, y.data_ptr(), stat.data_ptr(), n, h, w, cin, cout, 3, 3, 1, 1,
                                  int(reflect), 1, h, w, 0, st)
            assert rc == 0, L.vqseg_last_error()
            torch.cuda.synchronize()
            res[wide] = (y.cpu(), stat.cpu())
        finally:
            _hip.set_option("conv3x3_patch_wide_tile", prev)
    assert rel(res[1][0].float(), ref) < 2 ** -7
    assert torch.equal(res[1][0].view(torch.int16), res[0][0].view(torch.int16))
    assert torch.equal(res[1][1], res[0][1])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_residual_batchnorm_backward_with_the_masked_gradient_written_by_the_reduce_pass(dtype):
    """r4: for a layer with a residual branch the backward's reduce pass writes the ReLU-masked gradient (= the residual branch's
    gradient) and the apply pass reads that instead of (g_out, out) -- one tensor read less; in bf16 the forward also leaves the
    ReLU mask as a bit field and the reduce pass reads that instead of `out` (vqseg_bn_apply_bits_f / vqseg_bn_backward_bits_f).
    Every output (activation, input / residual / weight / BatchNorm gradients) bit-identical between the three forms: bit field,
    masked gradient from `out` (py_bn_bits = 0), r3's two passes over g_out and out (bn_bwd_premask = 0)."""
    import copy
    from vq_seg_amd import _hip, nnf
    torch.manual_seed(2)
    conv = nn.Conv2d(64, 128, 1, bias=False).to(dev())
    bn = nn.BatchNorm2d(128).to(dev())
    x = cl(synth.uniform(1, (3, 64, 12, 20), -1, 1)).to(dev())
    r = cl(synth.uniform(2, (3, 128, 12, 20), -1, 1)).to(dev())
    g = cl(synth.uniform(3, (3, 128, 12, 20), -1, 1)).to(dev())
    res = {}
    nnf.lib()
    for bits, opt in ((1, 1), (0, 1), (0, 0)):
        prev = _hip.set_option("bn_bwd_premask", opt)
        prev_bits = _hip.PY_OPTS.get("py_bn_bits")
        _hip.PY_OPTS["py_bn_bits"] = bits
        try:
            c, b = copy.deepcopy(conv), copy.deepcopy(bn)
            xx, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
            if dtype == torch.bfloat16:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = nnf.conv_bn_act(xx.to(dtype), c, b, relu=True, residual=rr.to(dtype))
            else:
                out = nnf.conv_bn_act(xx, c, b, relu=True, residual=rr)
            assert out.grad_fn.mask_bits == (bits == 1 and dtype == torch.bfloat16)
            out.backward(g.to(out.dtype))
            torch.cuda.synchronize()
            res[bits, opt] = (out.detach().float(), xx.grad, rr.grad, c.weight.grad, b.weight.grad, b.bias.grad)
        finally:
            _hip.set_option("bn_bwd_premask", prev)
            if prev_bits is None:
                _hip.PY_OPTS.pop("py_bn_bits", None)
            else:
                _hip.PY_OPTS["py_bn_bits"] = prev_bits
    for key in ((1, 1), (0, 1)):
        for i, (a, b_) in enumerate(zip(res[key], res[0, 0])):
            assert torch.equal(a, b_), (key, i)


@pytest.mark.gpu
def test_batchnorm_apply_writes_the_relu_mask_as_a_bit_field():
    """vqseg_bn_apply_bits_f: out as vqseg_bn_apply_f's; bit (i % 8) of bits[i / 8] = (out[i] > 0) over the flat index -- tested
    on the value as STORED (bf16), incl. pre-activations that round to zero; rejects channel counts that are not multiples of 8."""
    from vq_seg_amd import _hip, nnf
    L = nnf.lib()
    m, c = 1000, 72
    y = synth.uniform(5, (m, c), -2, 2).to(dev()).to(torch.bfloat16)
    r = synth.uniform(6, (m, c), -2, 2).to(dev()).to(torch.bfloat16)
    y[:, 0] = 1e-30                                                          # 1e-30 * 1e-12 is positive in fp32, below bf16's smallest subnormal
    r[:, 0] = 0
    sc = synth.uniform(7, (c,), 0.5, 1.5).to(dev())
    sh = synth.uniform(8, (c,), -0.5, 0.5).to(dev())
    sc[0], sh[0] = 1e-12, 0.0
    out = torch.empty_like(y)
    ref = torch.empty_like(y)
    bits = torch.zeros(m * c // 8, dtype=torch.uint8, device=dev())
    s = torch.cuda.current_stream().cuda_stream
    assert L.vqseg_bn_apply_bits_f(y.data_ptr(), r.data_ptr(), sc.data_ptr(), sh.data_ptr(), m, c, out.data_ptr(), bits.data_ptr(), s) == 0
    assert L.vqseg_bn_apply_f(1, y.data_ptr(), r.data_ptr(), sc.data_ptr(), sh.data_ptr(), m, c, 1, ref.data_ptr(), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    want = (out.reshape(-1, 8) > 0).to(torch.int32)
    want = (want << torch.arange(8, device=dev(), dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert not bool((out[:, 0] > 0).any())
    assert L.vqseg_bn_apply_bits_f(y.data_ptr(), r.data_ptr(), sc.data_ptr(), sh.data_ptr(), m * 2, c // 2, out.data_ptr(), bits.data_ptr(), s) != 0


@pytest.mark.gpu
def test_bottleneck_shortcut_gradient_masked_by_the_first_convs_epilogue():
    """r4: in an identity Bottleneck (bf16) the shortcut gradient g_out .* (out > 0) is never stored: bn3's backward hands (g_out,
    mask bits) over the GradLink and conv1's data-gradient epilogue masks while adding (vqseg_conv2d_affine_bits_f).  Every
    gradient bit-identical to the stored form (py_gres_bits = 0) and to the form without bit fields (py_bn_bits = 0); the stock-
    operator fallback (_mask_with_bits) agrees with (out > 0)."""
    import copy
    from vq_seg_amd import _hip, nnf
    from vq_seg_amd.models.encoders.resnet import Bottleneck
    torch.manual_seed(4)
    blk0 = Bottleneck(256, 64).to(dev()).train()
    x = cl(synth.uniform(1, (3, 256, 20, 12), -1, 1)).to(dev())
    g = cl(synth.uniform(2, (3, 256, 20, 12), -1, 1)).to(dev())
    nnf.lib()
    res = {}
    for key in ((1, 1), (1, 0), (0, 0)):
        saved = {k: _hip.PY_OPTS.get(k) for k in ("py_bn_bits", "py_gres_bits")}
        _hip.PY_OPTS["py_bn_bits"], _hip.PY_OPTS["py_gres_bits"] = key
        try:
            blk = copy.deepcopy(blk0)
            xx = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = blk(xx.to(torch.bfloat16) * 1.0)
            out.backward(g.to(out.dtype))
            torch.cuda.synchronize()
            res[key] = [out.detach().float(), xx.grad] + [p.grad for p in blk.parameters()]
        finally:
            for k, v in saved.items():
                if v is None:
                    _hip.PY_OPTS.pop(k, None)
                else:
                    _hip.PY_OPTS[k] = v
    for key in ((1, 1), (1, 0)):
        for i, (a, b_) in enumerate(zip(res[key], res[0, 0])):
            assert torch.equal(a, b_), (key, i)
    o = synth.uniform(3, (5, 7, 9, 16), -1, 1).to(dev()).to(torch.bfloat16)
    gg = synth.uniform(4, (5, 7, 9, 16), -1, 1).to(dev()).to(torch.bfloat16)
    keep = (o.reshape(-1, 8) > 0).to(torch.int32)
    bits = (keep << torch.arange(8, device=dev(), dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(nnf._mask_with_bits(gg, bits), torch.where(o > 0, gg, torch.zeros_like(gg)))


@pytest.mark.gpu
@pytest.mark.parametrize("k,cin,cout,hw", [(1, 64, 256, (20, 12)), (3, 64, 64, (18, 22)), (3, 128, 256, (9, 16)), (1, 256, 72, (7, 5))])
def test_conv_epilogue_masks_the_addend_with_a_bit_field(k, cin, cout, hw):
    """vqseg_conv2d_affine_bits_f (y = conv + res .* bits) against vqseg_conv2d_affine_f on the pre-masked addend: bit-identical,
    on the 1x1 kernels and the 3x3 patch kernel (one shared epilogue), incl. a channel count that is a multiple of 8 only and pixel
    counts that end inside a tile; rejects Cout % 8 != 0 and null bit fields."""
    from vq_seg_amd import nnf
    L = nnf.lib()
    h, w = hw
    n = 3
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=False).to(dev())
    w_hi, _ = nnf.packed_weights(conv.weight, False, False)
    x = synth.uniform(1, (n, h, w, cin), -1, 1).to(dev()).to(torch.bfloat16)
    res = synth.uniform(2, (n, h, w, cout), -1, 1).to(dev()).to(torch.bfloat16)
    o = synth.uniform(3, (n, h, w, cout), -1, 1).to(dev())
    keep = (o.reshape(-1, 8) > 0).to(torch.int32)
    bits = (keep << torch.arange(8, device=dev(), dtype=torch.int32)).sum(1).to(torch.uint8)
    masked = torch.where(o > 0, res, torch.zeros_like(res))
    one, zero = torch.ones(cout, device=dev()), torch.zeros(cout, device=dev())
    y0, y1 = torch.empty_like(res), torch.empty_like(res)
    s = torch.cuda.current_stream().cuda_stream
    assert L.vqseg_conv2d_affine_f(x.data_ptr(), None, cin, w_hi.data_ptr(), None, one.data_ptr(), zero.data_ptr(), masked.data_ptr(), 0,
                                   y0.data_ptr(), n, h, w, cin, cout, k, k, 1, k // 2, 0, h, w, 0, s) == 0, L.vqseg_last_error()
    assert L.vqseg_conv2d_affine_bits_f(x.data_ptr(), w_hi.data_ptr(), one.data_ptr(), zero.data_ptr(), res.data_ptr(), bits.data_ptr(),
                                        y1.data_ptr(), n, h, w, cin, cout, k, k, k // 2, h, w, s) == 0, L.vqseg_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert bool((y0.float() != 0).any())
    assert L.vqseg_conv2d_affine_bits_f(x.data_ptr(), w_hi.data_ptr(), one.data_ptr(), zero.data_ptr(), res.data_ptr(), None,
                                        y1.data_ptr(), n, h, w, cin, cout, k, k, k // 2, h, w, s) != 0
    assert L.vqseg_conv2d_affine_bits_f(x.data_ptr(), w_hi.data_ptr(), one.data_ptr(), zero.data_ptr(), res.data_ptr(), bits.data_ptr(),
                                        y1.data_ptr(), n, h, w, cin, cout - 4, k, k, k // 2, h, w, s) != 0
