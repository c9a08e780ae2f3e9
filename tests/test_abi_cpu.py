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
