"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, validates arguments, and refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

from vq_seg_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vqseg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqseg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _hip.lib()
    declared = header_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vqseg.h but not exported"
        assert name in _hip.SYMBOLS, f"{name} has no ctypes signature in vq_seg_amd/_hip.py"
    assert sorted(_hip.SYMBOLS) == declared
    assert L.vqseg_abi_version() == 1


def test_workspace_and_argument_validation_without_gpu():
    L = _hip.lib()
    assert L.vqseg_vq_workspace_bytes(8192, 512, 512) >= 512 * 512 * 4
    assert L.vqseg_vq_workspace_bytes(0, 512, 512) == 0
    assert L.vqseg_kmeans_workspace_bytes(8192, 512, 512) > L.vqseg_vq_workspace_bytes(8192, 512, 512)
    # null pointers / bad shapes are rejected before anything touches a device
    rc = L.vqseg_vq_assign_f32(None, None, None, 16, 6, 8, None, None, None, 0, None)
    assert rc == -1 and b"multiple of 4" in L.vqseg_last_error()
    rc = L.vqseg_vq_assign_f32(None, None, None, 16, 8, 8, None, None, None, 0, None)
    assert rc == -1 and b"null" in L.vqseg_last_error()
    assert L.vqseg_kernel_name(b"vqseg_vq_forward_f32") == b"vq_assign_f32_kernel"


def test_no_cpu_fallback():
    from vq_seg_amd.vector_quantizer import VectorQuantizer
    vq = VectorQuantizer(dim=16, num_embeddings=8)
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        vq(torch.rand(1, 16, 4, 4))


def test_make_vq_module_contract():
    from vq_seg_amd.vector_quantizer import Identity, VectorQuantizer, make_vq_module
    enc = (3, 64, 256, 512, 1024, 2048)
    cfg = {"num_embeddings": [0, 0, 512, 512, 512], "distance": "euclidean", "kmeans_init": True}
    mods = make_vq_module(cfg, enc, 5)
    assert [type(m) for m in mods] == [Identity, Identity, VectorQuantizer, VectorQuantizer, VectorQuantizer]
    assert mods[2].codebook.embedding.weight.shape == (512, 512)
    assert mods[4].codebook.embedding.weight.shape == (512, 2048)
    assert mods[2].codebook.initted is False                     # kmeans_init -> N(0,1) until first train forward
    assert list(mods.state_dict()) == [f"{i}.codebook.embedding.weight" for i in (2, 3, 4)]
    x = torch.rand(1, 4, 2, 2)
    assert mods[0](x)[0] is x and mods[0](x)[1:] == (None, None, None)
    every = make_vq_module({"num_embeddings": 16}, enc, 5)
    assert all(isinstance(m, VectorQuantizer) for m in every) and every[0].codebook.initted
    w = every[0].codebook.embedding.weight
    assert w.abs().max() <= 1 / 16                               # U(-1/K, 1/K), vq_img.py:156-158
    with pytest.raises(ValueError):
        make_vq_module({"num_embeddings": [0, 0, -1, 4, 4]}, enc, 5)
    with pytest.raises(TypeError):
        make_vq_module({"num_embeddings": "512"}, enc, 5)
    with pytest.raises(AssertionError):
        make_vq_module({"num_embeddings": [0, 512]}, enc, 5)
    with pytest.raises(NotImplementedError):
        VectorQuantizer(dim=8, num_embeddings=4, distance="cosine")


# ---------------------------------------------------------------------------------------------------------------------------
# The ctypes boundary takes untyped pointers: every tensor crosses it through _hip.tptr, which checks the element type against
# the flag / entry point it travels with, density, the element count the sizes imply, and the device -- type and size BEFORE the
# device, so a box without a GPU can feed every wrapper the wrong thing (the r2 fault was a bf16 buffer
# behind the f32 entry point).
# ---------------------------------------------------------------------------------------------------------------------------
def test_tptr_checks_type_flag_size_density_then_device():
    E = _hip.HipLibraryError
    f32, bf = torch.zeros(4, 8), torch.zeros(4, 8, dtype=torch.bfloat16)
    assert _hip.tptr(None, "x") is None
    with pytest.raises(E, match="expected torch.bfloat16, got torch.float32"):
        _hip.tptr(f32, "rows", bf=1)
    with pytest.raises(E, match="expected torch.float32, got torch.bfloat16"):
        _hip.tptr(bf, "rows", bf=0)
    with pytest.raises(E, match="expected torch.bfloat16"):
        _hip.tptr(f32, "split-3 rows", bf=2)
    with pytest.raises(E, match="expected torch.int16"):
        _hip.tptr(bf, "packed weights", dtype=torch.int16)
    with pytest.raises(E, match="need 64 elements, the tensor has 32"):
        _hip.tptr(f32, "rows", bf=0, numel=64)
    with pytest.raises(E, match="at least 33"):
        _hip.tptr(f32, "rows", bf=0, numel=33, at_least=True)
    with pytest.raises(E, match="dense"):
        _hip.tptr(f32[:, ::2], "rows", bf=0)
    with pytest.raises(E, match="expected a tensor"):
        _hip.tptr([1.0], "rows")
    # everything else right: only now the device matters -- remembered, and raised by _stream(), the last argument of every launch,
    # so that the other tensors of the same call get their type / size checks first
    assert _hip.tptr(f32, "rows", bf=0, numel=32) == 0
    with pytest.raises(E, match="rows: the HIP path needs .* no CPU fallback"):
        _hip._stream()
    assert _hip.tptr(f32, "rows", bf=0) == 0
    with pytest.raises(E, match="expected torch.int64"):            # a later type error wins and clears the pending device error
        _hip.tptr(f32, "idx", dtype=torch.int64)
    assert not _hip._NOT_ON_GPU


WRONG = [
    # (what, call)  -- CPU tensors of the WRONG type / size into each _hip wrapper: the error must name the type / size, not the device
    ("vq_forward f64 rows", lambda: _hip.vq_forward(torch.zeros(8, 16, dtype=torch.float64), torch.zeros(4, 16), False, 1.0), "expected torch.float32"),
    ("vq_forward f16 rows", lambda: _hip.vq_forward(torch.zeros(8, 16, dtype=torch.float16), torch.zeros(4, 16), False, 1.0), "expected torch.float32"),
    ("vq_forward bf16 codebook", lambda: _hip.vq_forward(torch.zeros(8, 16), torch.zeros(4, 16, dtype=torch.bfloat16), False, 1.0), "codebook: expected torch.float32"),
    ("vq_forward codebook width", lambda: _hip.vq_forward(torch.zeros(8, 16), torch.zeros(4, 32), False, 1.0), "codebook: the sizes passed along need 64"),
    ("vq_forward prepared blob", lambda: _hip.vq_forward(torch.zeros(8, 16), torch.zeros(4, 16), False, 1.0, prepared=torch.zeros(3, dtype=torch.uint8)),
     "prepared codebook: the sizes"),
    ("vq_assign f16 rows", lambda: _hip.vq_assign(torch.zeros(8, 16, dtype=torch.float16), torch.zeros(4, 16)), "expected torch.float32"),
    ("vq_assign strided rows", lambda: _hip.vq_assign(torch.zeros(8, 32)[:, ::2], torch.zeros(4, 16)), "dense"),
    ("vq_forward_group mixed", lambda: _hip.vq_forward_group([torch.zeros(8, 16), torch.zeros(8, 16, dtype=torch.bfloat16)], [torch.zeros(4, 16)] * 2,
                                                             [None, None], False, [1.0, 1.0]), "one row type per call"),
    ("vq_backward bf16 grad", lambda: _hip.vq_backward(torch.zeros(8, 16, dtype=torch.bfloat16), None, torch.zeros(8, 16), torch.zeros(8, 16), 1.0),
     "grad_quant: expected torch.float32"),
    ("vq_backward size", lambda: _hip.vq_backward(torch.zeros(4, 16), None, torch.zeros(8, 16), torch.zeros(8, 16), 1.0), "grad_quant: the sizes"),
    ("vq_backward_bf16 f32 rows", lambda: _hip.vq_backward_bf16(torch.zeros(8, 16, dtype=torch.bfloat16), None, torch.zeros(8, 16),
                                                                torch.zeros(8, dtype=torch.int64), torch.zeros(4, 16), 1.0), "rows: expected torch.bfloat16"),
    ("vq_backward_bf16 i32 idx", lambda: _hip.vq_backward_bf16(torch.zeros(8, 16, dtype=torch.bfloat16), None, torch.zeros(8, 16, dtype=torch.bfloat16),
                                                               torch.zeros(8, dtype=torch.int32), torch.zeros(4, 16), 1.0), "idx: expected torch.int64"),
    ("kmeans bf16 samples", lambda: _hip.kmeans(torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(4, 16), 2), "samples: expected torch.float32"),
    ("kmeans means width", lambda: _hip.kmeans(torch.zeros(8, 16), torch.zeros(4, 8), 2), "means: the sizes"),
    ("kmeans_accumulate", lambda: _hip.kmeans_accumulate(torch.zeros(8, 16), torch.zeros(4, 16, dtype=torch.float64)), "means: expected torch.float32"),
    ("kmeans_finalize i32 counts", lambda: _hip.kmeans_finalize(torch.zeros(4, 16), torch.zeros(4, dtype=torch.int32), torch.zeros(4, 16)), "counts: expected torch.int64"),
    ("vq_code_sums i32 idx", lambda: _hip.vq_code_sums(torch.zeros(8, 16), torch.zeros(8, dtype=torch.int32), 4), "idx: expected torch.int64"),
    ("vq_ema_update", lambda: _hip.vq_ema_update(torch.zeros(4), torch.zeros(4, 16), torch.zeros(4, 16, dtype=torch.bfloat16), torch.zeros(4, 16),
                                                 torch.zeros(4, dtype=torch.int64), 0.9, 1e-5), "codebook: expected torch.float32"),
    ("vq_prepare", lambda: _hip.vq_prepare(torch.zeros(4, 16, dtype=torch.bfloat16)), "codebook: expected torch.float32"),
]


@pytest.mark.parametrize("what,call,msg", WRONG, ids=[w[0] for w in WRONG])
def test_hip_wrappers_refuse_wrong_types_and_sizes_before_touching_a_device(what, call, msg):
    with pytest.raises((_hip.HipLibraryError, ValueError), match=msg):
        call()


def test_nnf_kernels_refuse_wrong_types_before_touching_a_device():
    """The autograd functions behind nnf.* hand their tensors over through the same helper: wrong element types raise in Python."""
    from vq_seg_amd import nnf
    E = _hip.HipLibraryError
    x64 = torch.zeros(1, 8, 4, 4, dtype=torch.float64)
    for fn in (lambda: nnf._MaxPool.apply(x64), lambda: nnf._Bilinear.apply(x64, 8, 8, False),
               lambda: nnf._Head1x1.apply(x64, torch.zeros(3, 8, 1, 1))):
        with pytest.raises(E, match="float32 .* or bfloat16"):
            fn()
    xb = torch.zeros(1, 8, 4, 4, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with pytest.raises(E, match="head weight: expected torch.float32"):
        nnf._Head1x1.apply(xb, torch.zeros(3, 8, 1, 1, dtype=torch.bfloat16))
    with pytest.raises(E, match="split-3 rows: expected torch.bfloat16"):   # an S3 object whose rows are not bf16
        nnf.S3(torch.zeros(1, 4, 4, 16), 8).float()
    with pytest.raises(E, match="need 256 elements"):               # ... or whose channel count disagrees with its rows
        nnf.S3(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16), 8).float()
    with pytest.raises(E, match="logits: expected a float32"):
        nnf._strided_f32(torch.zeros(2, 3, 4, 4, dtype=torch.bfloat16), "logits", 96)
    with pytest.raises(E, match="logits: the sizes passed along need 96"):
        nnf._strided_f32(torch.zeros(1, 3, 4, 4), "logits", 96)
    with pytest.raises(E, match="no CPU fallback"):
        nnf.softmax_stats(torch.zeros(1, 3, 4, 4), True, True)       # all types right on a CPU tensor: the device error surfaces


def test_hip_adam_has_no_cpu_path_and_keeps_torch_adams_layout():
    """optim.HipAdam (r4): a torch.optim.Adam whose step is the HIP kernel -- CPU parameters are refused (no fallback), the
    argument checks of the entry point and the work-item count are host-side."""
    from vq_seg_amd.optim import HipAdam
    p = torch.nn.Parameter(torch.ones(4, 4))
    opt = HipAdam([p], lr=0.1)
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["betas"] == (0.9, 0.999) and opt.defaults["eps"] == 1e-8
    p.grad = torch.ones(4, 4)
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        opt.step()
    assert torch.equal(p.detach(), torch.ones(4, 4))              # nothing was updated behind the error
    L = _hip.lib()
    assert L.vqseg_adam_work_items(4096 * 3 + 1, 0, 0, 0) == 4
    assert L.vqseg_adam_work_items(64 * 32 * 9, 3, 64, 32) == 2 and L.vqseg_adam_work_items(256 * 64, 1, 256, 64) == 8
    assert L.vqseg_adam_step_f32(None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 1, None) == -1
    # the second source of a two-use weight gradient must come with its tensors
    assert L.vqseg_conv2d_wgrad2_f(None, None, None, 1, None, None, None, 1, 8, 4, 4, 8, 4, 4, 8, 3, 3, 1, 1, 0, 0, 8, 0, 0, None, 0, None, None) == -1


def test_a_failed_fused_batchnorm_launch_zeroes_the_modules_counters():
    """the opt-in fused BatchNorm launches hand over through per-module counters that must read zero between launches; an
    entry point that returns an error may leave them dirty -- the host wrapper zeroes them before raising."""
    import torch
    from torch import nn
    from vq_seg_amd import _hip, nnf
    bn = nn.BatchNorm2d(8)
    bn._vq_sync = torch.tensor([3, 0, 1, 0], dtype=torch.int32)
    with pytest.raises(_hip.HipLibraryError):
        nnf._check_fused_bn(_hip.lib().vqseg_bn_apply_f(1, None, None, None, None, 0, 0, 0, None, None), "vqseg_bn_apply_f (null arguments)", bn)
    assert int(bn._vq_sync.abs().sum()) == 0
    nnf._check_fused_bn(0, "ok", bn)                            # a clean return code touches nothing
