"""The split-3 eval forward (nnf.S3: fp32-precision no-grad eval passes on the bf16 LDS-DMA / patch-reuse kernels) against
(1) the CPU fp32 oracle per operator, (2) the reference's eval logits (fixture model_v1 / model_v2, 1e-3 of scale: north_star),
(3) the register-staged "precise" kernels it replaces (VQSEG_OPTS py_s3_eval=0 path) -- code indices identical."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from tests import cases, golden_io, synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_split_merge_round_trip_and_elementwise_ops():
    from vq_seg_amd import nnf
    x = (synth.uniform(1, (2, 64, 12, 10), -3, 3) * synth.uniform(2, (2, 64, 12, 10), 0, 1) ** 4).to(dev())     # wide dynamic range
    s = nnf.to_s3(x.contiguous(memory_format=torch.channels_last))
    assert s.shape == x.shape and s.rows.shape == (2, 12, 10, 128) and s.rows.dtype == torch.bfloat16      # [hi | lo]
    back = s.float()
    assert ((back - x).abs() <= 2.0 ** -16 * x.abs() + 1e-30).all()                       # hi + lo keeps ~17 bits
    assert torch.equal(s.rows[..., :64], x.permute(0, 2, 3, 1).to(torch.bfloat16))                 # hi = bf16(v)
    assert rel(nnf.max_pool_3x3_s2(s).float(), F.max_pool2d(back, 3, 2, 1)) < 1e-5
    for size, align in (((24, 20), False), ((17, 23), False), ((24, 20), True)):
        got = nnf.upsample_bilinear(s, size=size, align_corners=align).float()
        assert rel(got, F.interpolate(back, size=size, mode="bilinear", align_corners=align)) < 1e-5


@pytest.mark.parametrize("cin,c2,cout,k,stride,reflect,hw,res", [
    (64, 0, 64, 1, 1, False, 24, False), (64, 0, 256, 1, 1, False, 24, True), (128, 0, 128, 3, 1, True, 32, False),
    (128, 0, 128, 3, 2, True, 32, False), (256, 0, 512, 1, 2, False, 16, False), (128, 64, 32, 3, 1, False, 32, False),
    (32, 0, 32, 3, 1, False, 32, False), (256, 256, 128, 3, 1, False, 16, False), (512, 0, 128, 1, 1, False, 8, True)])
def test_conv_bn_act_split3_matches_cpu_fp32(cin, c2, cout, k, stride, reflect, hw, res):
    """Every layer shape class of the network (1x1, 3x3 zero / reflect, stride 2, concat, residual, narrow) in split-3 form
    against nn.Conv2d + eval BatchNorm2d (+ residual) + ReLU on the CPU in fp32: 2e-5 of scale (the precise mode's bar is 2e-4)."""
    from vq_seg_amd import nnf, _hip
    torch.manual_seed(cin + cout + k)
    n = 8
    conv = nn.Conv2d(cin + c2, cout, k, stride, k // 2, bias=False, padding_mode="reflect" if reflect else "zeros")
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5), bn.running_mean.uniform_(-0.2, 0.2), bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    x = synth.relu_features(3, (n, cin, hw, hw))
    x2 = synth.relu_features(4, (n, c2, hw, hw)) if c2 else None
    ho = (hw + 2 * (k // 2) - k) // stride + 1
    r = synth.uniform(5, (n, cout, ho, ho), -1, 1) if res else None
    with torch.no_grad():
        want = bn(conv(torch.cat([x, x2], 1) if c2 else x))
        want = F.relu(want + r) if res else F.relu(want)
    conv, bn = conv.to(dev()), bn.to(dev())
    cl = lambda t: t.to(dev()).contiguous(memory_format=torch.channels_last)
    L = _hip.lib()
    prev_min = L.vqseg_set_option(b"conv3x3_patch_min_workgroups", 1)                     # small test shapes reach the patch kernels too,
    prev_512 = L.vqseg_set_option(b"conv3x3_patch_tile512_min_workgroups", 1)             # ... and their 512-pixel tiles (32 / 64 outputs)
    L.vqseg_set_option(b"conv3x3_patch_tile512_launches", 0)
    try:
        with torch.no_grad():
            got = nnf.conv_bn_act(nnf.to_s3(cl(x)), conv, bn, relu=True, x2=nnf.to_s3(cl(x2)) if c2 else None,
                                  residual=nnf.to_s3(cl(r)) if res else None)
    finally:
        L.vqseg_set_option(b"conv3x3_patch_min_workgroups", prev_min)
        L.vqseg_set_option(b"conv3x3_patch_tile512_min_workgroups", prev_512)
    tall = L.vqseg_set_option(b"conv3x3_patch_tile512_launches", 0)
    assert (tall >= 1) == (k == 3 and stride == 1 and cout == 32 and hw % 32 == 0), "512-pixel split-3 tile dispatch"
    assert isinstance(got, nnf.S3) and got.shape == want.shape
    assert rel(got.float(), want) < 2e-5


@pytest.mark.parametrize("version", [1, 2])
def test_whole_model_eval_forward_split3_vs_reference_and_vs_precise_kernels(version):
    from tests.test_model_gpu import build
    from vq_seg_amd import _hip, nnf
    fx = golden_io.load(f"model_v{version}")
    model = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"])
    x, _gt, _ = cases.model_inputs()
    x = x.to(dev())
    model.eval()
    outs = {}
    for s3 in (1, 0):
        _hip.lib()
        _hip.PY_OPTS["py_s3_eval"] = s3
        try:
            with torch.no_grad():
                logits, closs, usage, proto = model(x)
                feats, _l, _u = model.quantize(model.encode(x))
            outs[s3] = (logits, usage, [type(f).__name__ for f in feats])
        finally:
            _hip.PY_OPTS.pop("py_s3_eval", None)
    assert "S3" in outs[1][2] and "S3" not in outs[0][2]                                   # the split-3 path really ran / really did not
    for s3 in (1, 0):
        err = rel(outs[s3][0], fx["eval_logits"])
        print(f"v{version} eval logits vs reference, split3={s3}: {err:.2e} of scale")
        assert err <= 1e-3
        assert torch.allclose(outs[s3][1].double(), fx["eval_usage"].double(), rtol=1e-6)  # dead-code % exact -> same code histogram
    assert rel(outs[1][0], outs[0][0]) < 2e-4


def test_split3_is_confined_to_the_model_forward():
    """A bare `model.encoder(x)` (no scope) returns plain fp32 tensors; training / grad-enabled / autocast forwards never split."""
    from tests.test_model_gpu import build
    from vq_seg_amd import nnf
    fx = golden_io.load("model_v1")
    model = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"])
    x, gt, _ = cases.model_inputs()
    x, gt = x.to(dev()), gt.to(dev())
    model.eval()
    with torch.no_grad():
        feats = model.encoder(x.contiguous(memory_format=torch.channels_last))
        assert all(isinstance(f, torch.Tensor) for f in feats)
        assert any(isinstance(f, nnf.S3) for f in model.encode(x))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert not any(isinstance(f, nnf.S3) for f in model.encode(x))
    assert not any(isinstance(f, nnf.S3) for f in model.encode(x))                         # grad enabled
    model.train()
    with torch.no_grad():
        assert not any(isinstance(f, nnf.S3) for f in model.encode(x))


def test_eval_forward_with_widths_the_split3_kernels_reject_falls_back_to_the_precise_path():
    """the split-3 kernels need 32-channel K chunks per concat segment and Cout % 8 == 0.  A model whose decoder
    ends in 16 channels (the next layer then has Cin = 16) must still run its no-grad eval forward: those layers merge back to fp32
    and take the precise kernels.  Logits equal the all-precise forward (py_s3_eval = 0) to split-3 accuracy; indices identical."""
    from vq_seg_amd import _hip
    from vq_seg_amd.models.networks import make_model
    cfg = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5, "decoder_channels": [256, 128, 64, 32, 16],
                                               "vq_cfg": {"num_embeddings": [0, 0, 32, 32, 32], "distance": "euclidean", "kmeans_init": False},
                                               "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    torch.manual_seed(3)
    try:
        model = make_model(cfg).to(dev())
    except TypeError:                                                   # the factory takes the reference's ctor kwargs only
        cfg["params"].pop("decoder_channels")
        model = make_model(cfg).to(dev())
        from vq_seg_amd.models.networks.unet.decoder import UnetDecoder
        model.decoder = UnetDecoder((3, 64, 256, 512, 1024, 2048), [256, 128, 64, 32, 16]).to(dev())
        model.segmentation_head = nn.Conv2d(16, 3, 1, bias=False).to(dev())
    model.eval()
    x = synth.uniform(9, (2, 3, 64, 64)).to(dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = model(x)[0]
        _hip.lib()
        prev = _hip.PY_OPTS.get("py_s3_eval")
        _hip.PY_OPTS["py_s3_eval"] = 0
        try:
            want = model(x)[0]
        finally:
            if prev is None:
                _hip.PY_OPTS.pop("py_s3_eval")
            else:
                _hip.PY_OPTS["py_s3_eval"] = prev
    assert got.shape == want.shape == (2, 3, 64, 64) and torch.isfinite(got).all()
    assert rel(got, want) < 2e-4, rel(got, want)


def test_split3_exact_2x_resize_is_bit_identical_to_the_generic_kernel():
    """s3_bilinear_up2_kernel (2x, align_corners = False, power-of-two extents) against s3_bilinear_kernel (`bilinear_up2` = 0)."""
    from vq_seg_amd import _hip, nnf
    L = _hip.lib()
    for shape in ((2, 64, 16, 16), (1, 8, 1, 2), (2, 256, 4, 32)):
        x = (synth.uniform(5, shape, -3, 3) * synth.uniform(6, shape, 0, 1) ** 3).to(dev()).contiguous(memory_format=torch.channels_last)
        s = nnf.to_s3(x)
        size = (2 * shape[2], 2 * shape[3])
        fast = nnf.upsample_bilinear(s, size=size, align_corners=False).rows.clone()
        prev = L.vqseg_set_option(b"bilinear_up2", 0)
        try:
            generic = nnf.upsample_bilinear(s, size=size, align_corners=False).rows.clone()
        finally:
            L.vqseg_set_option(b"bilinear_up2", prev)
        assert torch.equal(fast, generic)


@pytest.mark.parametrize("res", [False, True])
def test_split3_conv_on_the_256_channel_tile_is_bit_identical_to_the_128_channel_tile(res):
    """r4: the split-3 3x3 convolution on the patch kernel's 256-channel tile (its [hi | lo] output leaves LDS in two column halves:
    both staging tiles of a 256 x 256 tile would need 270 KB) against the 128-channel tile: the same bits, with and without a
    residual; and against nn.Conv2d + eval BatchNorm2d (+ residual) + ReLU on the CPU in fp32 (2e-5 of scale)."""
    from vq_seg_amd import nnf, _hip
    torch.manual_seed(11)
    n, cin, cout, hw = 2, 64, 512, 32
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5), bn.running_mean.uniform_(-0.2, 0.2), bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    x = synth.relu_features(3, (n, cin, hw, hw))
    r = synth.uniform(5, (n, cout, hw, hw), -1, 1) if res else None
    with torch.no_grad():
        want = bn(conv(x))
        want = F.relu(want + r) if res else F.relu(want)
    conv, bn = conv.to(dev()), bn.to(dev())
    cl = lambda t: t.to(dev()).contiguous(memory_format=torch.channels_last)
    L = _hip.lib()
    prev_min = L.vqseg_set_option(b"conv3x3_patch_min_workgroups", 1)
    got = {}
    try:
        for wide in (1, 0):
            prev = L.vqseg_set_option(b"conv3x3_patch_wide_tile_s3", wide)
            try:
                with torch.no_grad():
                    got[wide] = nnf.conv_bn_act(nnf.to_s3(cl(x)), conv, bn, relu=True, residual=nnf.to_s3(cl(r)) if res else None)
            finally:
                L.vqseg_set_option(b"conv3x3_patch_wide_tile_s3", prev)
    finally:
        L.vqseg_set_option(b"conv3x3_patch_min_workgroups", prev_min)
    torch.cuda.synchronize()
    assert torch.equal(got[1].rows.view(torch.int16), got[0].rows.view(torch.int16))
    assert rel(got[1].float(), want) < 2e-5
