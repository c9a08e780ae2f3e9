"""Kernel-side parameter images (packed conv weights, prepared codebooks) must die when the parameter changes -- including the
changes the version counter does not see: a fused optimiser step and writes through `.data` (vq_seg_amd/_wcache.py)."""
import copy

import pytest
import torch
from torch import nn

from vq_seg_amd import _wcache


def test_optimizer_step_and_invalidate_drop_the_images_cpu():
    p, q = nn.Parameter(torch.ones(4)), nn.Parameter(torch.ones(4))
    for t in (p, q):
        _wcache.cache_of(t)["image"] = object()
    assert "image" in _wcache.cache_of(p)
    opt = torch.optim.SGD([p], lr=0.1)
    p.grad = torch.ones(4)
    opt.step()                                              # the global post-step hook
    assert "image" not in _wcache.cache_of(p) and "image" in _wcache.cache_of(q)
    q.data.copy_(torch.zeros(4))                            # invisible to the version counter ...
    assert "image" in _wcache.cache_of(q)
    _wcache.invalidate(q)                                   # ... hence the explicit call
    assert "image" not in _wcache.cache_of(q)
    m = nn.Linear(2, 2)
    _wcache.cache_of(m.weight)["image"] = 1
    _wcache.invalidate(m)
    assert "image" not in _wcache.cache_of(m.weight)
    _wcache.cache_of(m.weight)["image"] = 1
    with torch.no_grad():
        m.weight.mul_(2.0)                                  # ordinary in-place op: version counter
    assert "image" not in _wcache.cache_of(m.weight)


@pytest.mark.gpu
def test_fused_adam_step_reaches_the_conv_kernels():
    """torch.optim.Adam(fused=True) does not bump `_version`: after a step the convolution must still see the new weights
    (round 1 ran every step after the first on the first step's packed weights)."""
    from vq_seg_amd import nnf
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    conv, bn = nn.Conv2d(32, 64, 3, padding=1, bias=False).to(dev), nn.BatchNorm2d(64).to(dev)
    x = torch.rand(2, 32, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
    for dtype in (torch.float32, torch.bfloat16):
        xin = x.to(dtype)
        opt = torch.optim.Adam(list(conv.parameters()) + list(bn.parameters()), lr=0.05, fused=True)
        y0 = nnf.conv_bn_act(xin, conv, bn)
        y0.float().square().mean().backward()
        v = conv.weight._version
        opt.step()
        opt.zero_grad()
        y1 = nnf.conv_bn_act(xin, conv, bn)
        fresh_c, fresh_b = copy.deepcopy(conv), copy.deepcopy(bn)          # same values, no images
        assert getattr(fresh_c.weight, "_vq_pack", None) is None
        fresh_b.load_state_dict(bn.state_dict())
        bn2 = copy.deepcopy(bn)
        y_ref = nnf.conv_bn_act(xin, fresh_c, bn2)
        assert torch.equal(y1, y_ref), (dtype, conv.weight._version, v)
        assert not torch.equal(y1, y0)


@pytest.mark.gpu
def test_data_write_to_the_codebook_needs_invalidate_and_then_works():
    """`.data.copy_()` after a forward (the reference's own idiom, vq_img.py:185)."""
    from vq_seg_amd.vector_quantizer import VectorQuantizer
    from vq_seg_amd import nnf
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    vq = VectorQuantizer(dim=64, num_embeddings=32).to(dev).eval()
    x = torch.rand(2, 64, 8, 8, device=dev)
    new = torch.rand(32, 64, device=dev)
    idx0 = vq(x)[1]
    vq.codebook.embedding.weight.data.copy_(new)
    nnf.invalidate_weight_caches(vq)
    idx1 = vq(x)[1]
    ref = VectorQuantizer(dim=64, num_embeddings=32).to(dev).eval()
    with torch.no_grad():
        ref.codebook.embedding.weight.copy_(new)
    assert torch.equal(idx1, ref(x)[1]) and not torch.equal(idx1, idx0)
    sd = {k: v.clone() for k, v in vq.state_dict().items()}
    sd["codebook.embedding.weight"] = torch.rand(32, 64)
    vq.load_state_dict(sd)                                   # copy_ on the parameter: version counter, nothing to call
    ref.load_state_dict(sd)
    assert torch.equal(vq(x)[1], ref(x)[1])


@pytest.mark.gpu
@pytest.mark.parametrize("cout,cin,k,c1", [(64, 64, 3, 64), (128, 96, 3, 64), (40, 24, 3, 24), (256, 64, 1, 64), (32, 192, 3, 128), (1024, 2048, 3, 2048)])
def test_pack_all_is_bit_identical_to_the_single_image_packers(cout, cin, k, c1):
    """vqseg_conv_pack_all_f32 (one launch: forward, data-gradient and split-3 images) against the three single-image entry points."""
    from vq_seg_amd import _hip
    L = _hip.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(cout + cin)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    n_f, n_t = L.vqseg_conv_packed_elems(cout, cin, k, k, 0), L.vqseg_conv_packed_elems(cout, cin, k, k, 1)
    ref_f, ref_t = torch.full((n_f,), -1, dtype=torch.int16, device=dev), torch.full((n_t,), -1, dtype=torch.int16, device=dev)
    assert L.vqseg_conv_pack_weights_f32(w.data_ptr(), cout, cin, k, k, 0, ref_f.data_ptr(), None, st) == 0
    assert L.vqseg_conv_pack_weights_f32(w.data_ptr(), cout, cin, k, k, 1, ref_t.data_ptr(), None, st) == 0
    s3_ok = cin % 32 == 0 and c1 % 32 == 0
    ref_s = torch.full((cout * k * k * 3 * cin,), -1, dtype=torch.int16, device=dev)
    if s3_ok:
        assert L.vqseg_conv_pack_weights_s3_f32(w.data_ptr(), cout, cin, c1, k, k, ref_s.data_ptr(), st) == 0
    got_f, got_t, got_s = torch.full_like(ref_f, -2), torch.full_like(ref_t, -2), torch.full_like(ref_s, -2)
    assert L.vqseg_conv_pack_all_f32(w.data_ptr(), cout, cin, k, c1, got_f.data_ptr(), got_t.data_ptr(), got_s.data_ptr() if s3_ok else None, st) == 0
    assert torch.equal(got_f, ref_f) and torch.equal(got_t, ref_t)
    if s3_ok:
        assert torch.equal(got_s, ref_s)
