"""ctypes binding of libvqseg_hip.so (the C ABI declared in include/vqseg.h).

Torch is used only to own device memory and to name the current HIP stream; every
argument crossing the boundary is a raw pointer or a size.  Fails loudly: a missing
library, a CPU tensor, a wrong dtype or a non-zero return code raise.
"""
from __future__ import annotations

import ctypes
import os
import threading
import time
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQSEG_LIB") or os.path.join(_HERE, "libvqseg_hip.so")     # VQSEG_LIB: another build of the same ABI (kernel A/B runs)

# name -> (restype, argtypes); must list every symbol include/vqseg.h declares
SYMBOLS = {
    "vqseg_abi_version": (c_int, []),
    "vqseg_last_error": (c_char_p, []),
    "vqseg_kernel_name": (c_char_p, [c_char_p]),
    "vqseg_set_option": (c_int, [c_char_p, c_int]),
    "vqseg_profile_begin": (c_int, [c_int]),
    "vqseg_profile_collect": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_conv_profile_begin": (c_int, [c_int]),
    "vqseg_conv_profile_collect": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_vq_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vqseg_vq_filter_counter_offset": (c_size_t, [c_int64, c_int, c_int]),
    "vqseg_vq_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vqseg_vq_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vqseg_vq_forward_group": (c_int, [c_int, c_int] + [c_void_p] * 6 + [c_int, c_void_p] + [c_void_p] * 6 + [c_void_p]),
    "vqseg_vq_backward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "vqseg_vq_assign_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "vqseg_vq_assign_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "vqseg_vq_prepared_bytes": (c_size_t, [c_int, c_int]),
    "vqseg_vq_prepare_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vqseg_vq_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p,
                                      c_void_p]),
    "vqseg_kmeans_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vqseg_kmeans_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "vqseg_kmeans_accumulate_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                            c_size_t, c_void_p]),
    "vqseg_kmeans_finalize_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vqseg_vq_code_sums": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vqseg_vq_ema_update_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p,
                                        c_void_p]),
    "vqseg_conv_packed_elems": (c_size_t, [c_int] * 5),
    "vqseg_conv_pack_weights_f32": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p]),
    "vqseg_conv_stat_slots": (c_int64, [c_int64, c_int]),
    "vqseg_conv2d_f": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 14 + [c_void_p]),
    "vqseg_conv2d_affine_f": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p] +
                              [c_int] * 13 + [c_void_p]),
    "vqseg_conv2d_wgrad_workspace_bytes": (c_size_t, [c_int] * 9),
    "vqseg_conv2d_wgrad_f": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 17 + [c_void_p, c_size_t, c_void_p, c_void_p]),
    "vqseg_conv2d_wgrad2_f": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 16 +
                              [c_void_p, c_size_t, c_void_p, c_void_p]),
    "vqseg_bn_sync_ints": (c_int, [c_int]),
    "vqseg_bn_finalize_f": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_bn_apply_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "vqseg_bn_backward_workspace_floats": (c_size_t, [c_int64, c_int]),
    "vqseg_bn_backward_f": (c_int, [c_int] + [c_void_p] * 8 + [c_int64, c_int, c_int, c_int, c_int] + [c_void_p] * 7),
    "vqseg_conv2d_affine_bits_f": (c_int, [c_void_p] * 7 + [c_int] * 10 + [c_void_p]),
    "vqseg_bn_apply_bits_f": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vqseg_bn_backward_bits_f": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_int] + [c_void_p] * 7),
    "vqseg_maxpool3x3s2_f": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vqseg_bilinear_f": (c_int, [c_int, c_int, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "vqseg_head1x1_forward_f": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "vqseg_head1x1_backward_workspace_floats": (c_size_t, [c_int64, c_int, c_int]),
    "vqseg_head1x1_backward_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "vqseg_proto_loss_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vqseg_proto_loss_forward_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                           c_float, c_float, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "vqseg_proto_loss_backward_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                            c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vqseg_dice_workspace_bytes": (c_size_t, [c_int, c_int, c_int64]),
    "vqseg_dice_sums_forward_f": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_size_t,
                                          c_void_p, c_void_p, c_void_p]),
    "vqseg_dice_sums_backward_f": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "vqseg_dice_ce_sums_forward_f": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p,
                                             c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_dice_ce_sums_backward_f": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_softmax_stats_f": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_confusion_counts_f": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "vqseg_order_stats_workspace_bytes": (c_size_t, []),
    "vqseg_order_stats_f": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "vqseg_im2col_f": (c_int, [c_int, c_void_p] + [c_int] * 12 + [c_void_p, c_void_p]),
    "vqseg_reflect_fold_f": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqseg_reflect_ring_f": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vqseg_cast_f": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "vqseg_conv_pack_all_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_conv_packed_s2_elems": (c_size_t, [c_int, c_int, c_int]),
    "vqseg_conv_pack_weights_s2_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vqseg_conv2d_dgrad_s2_f": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "vqseg_stem7_conv_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "vqseg_conv2d_dgrad_s2_fold_rows": (ctypes.c_int64, [c_int] * 4),
    "vqseg_cps_loss_combine_f": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int, c_int,
                                         c_float, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqseg_conv2d_dgrad_s2_fold_f": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "vqseg_head1x1_backward_add_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "vqseg_maxpool3x3s2_backward_add_f": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqseg_conv_pack_weights_s3_f32": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_void_p]),
    "vqseg_s3_split_f": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "vqseg_s3_merge_f": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "vqseg_s3_maxpool3x3s2_f": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqseg_s3_bilinear_f": (c_int, [c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "vqseg_adam_work_items": (c_int64, [c_int64, c_int, c_int, c_int]),
    "vqseg_adam_step_f32": (c_int, [c_void_p, c_void_p, c_int, c_double, c_double, c_double, c_double, c_int64, c_void_p]),
}

_lib: Optional[ctypes.CDLL] = None


class HipLibraryError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C vq_seg_amd/csrc`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if handle.vqseg_abi_version() != 1:
            raise HipLibraryError("libvqseg_hip.so ABI version mismatch")
        for kv in filter(None, os.environ.get("VQSEG_OPTS", "").split(",")):     # dispatch tunables for A/B runs: "key=value,..."
            key, _, val = kv.partition("=")
            if key.strip().startswith("py_"):                # host-side switches (A/B runs of the Python layer): PY_OPTS
                PY_OPTS[key.strip()] = int(val)
                continue
            if handle.vqseg_set_option(key.strip().encode(), int(val)) < 0:
                raise HipLibraryError(f"VQSEG_OPTS: unknown option {key!r}")
        _lib = handle
    return _lib


PY_OPTS: dict = {}
FILTER_DIAG = None            # set to a list to collect (n, c, k, open-row counter) of every bf16 grouped VQ forward (diagnostics only)


def _check(rc: int, what: str) -> None:
    _raise_if_not_on_gpu()
    if rc != 0:
        msg = lib().vqseg_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what} failed (code {rc}): {msg}")


_BF_DTYPE = {0: torch.float32, 1: torch.bfloat16, 2: torch.bfloat16}      # the `bf16` flag of the entry points (2: split-3 [hi | lo] rows)


def tptr(t, name: str, dtype=None, numel: Optional[int] = None, bf: Optional[int] = None, at_least: bool = False) -> Optional[int]:
    """THE way a tensor crosses the C ABI: returns its address after checking what the untyped `void*` on the other side cannot --
    the element type (`dtype`: one torch dtype or a tuple; or `bf`: the 0 / 1 / 2 activation-type flag that is passed to the SAME
    entry point, so flag and buffer cannot disagree), dense memory, the element count the sizes passed along imply (`numel`, exact
    unless `at_least`), and a 'cuda' (ROCm) device.  A bf16 buffer behind an f32 flag is a 2x out-of-bounds read on the GPU (it
    happened: DESIGN 2, r2); here it is a Python exception.  The dtype / size checks come BEFORE the device check, so the CPU test
    suite can feed every wrapper the wrong type (tests/test_abi_cpu.py).  None passes through as a null pointer."""
    if t is None:
        return None
    if bf is not None:
        dtype = _BF_DTYPE[bf]
    try:                                                     # fast path: everything in order (one short-circuit expression per launch argument)
        if (dtype is None or t.dtype is dtype or (type(dtype) is tuple and t.dtype in dtype)) and t.is_cuda and \
                (numel is None or (t.numel() >= numel if at_least else t.numel() == numel)) and \
                (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
            return t.data_ptr()
    except AttributeError:
        pass
    return _tptr_report(t, name, dtype, numel, at_least)


def _tptr_report(t, name, dtype, numel, at_least):
    """slow path of tptr: say exactly what is wrong (type, density, size before the device)"""
    def bad(msg):
        _NOT_ON_GPU.clear()
        return HipLibraryError(f"{name}: {msg}")
    if not isinstance(t, torch.Tensor):
        raise bad(f"expected a tensor, got {type(t).__name__}")
    if dtype is not None and (t.dtype not in dtype if isinstance(dtype, tuple) else t.dtype != dtype):
        raise bad(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous() and not (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        raise bad(f"expected a dense (contiguous) tensor, got strides {tuple(t.stride())} for shape {tuple(t.shape)}")
    if numel is not None and (t.numel() < numel if at_least else t.numel() != numel):
        raise bad(f"the sizes passed along need {'at least ' if at_least else ''}{numel} elements, the tensor has {t.numel()} (shape {tuple(t.shape)})")
    if not t.is_cuda:
        # remembered, raised by _stream() -- the last argument every launch evaluates -- so that the OTHER tensors of the call still get
        # their type / size checks first (a box without a GPU can then exercise them: tests/test_abi_cpu.py); nothing is launched
        _NOT_ON_GPU.append(f"{name}: the HIP path needs a tensor on a 'cuda' (ROCm) device; got {t.device}. There is no CPU fallback.")
        return 0
    return t.data_ptr()


class _Pending(threading.local):
    """Deferred 'tensor is not on the GPU' messages of the wrapper call in progress -- per thread (autograd runs backward wrappers
    on its own threads), and short-lived: a message is raised by the SAME wrapper call microseconds later (_stream / _check); one
    that is still here after two seconds was orphaned by an unrelated exception between a tptr() and its launch and must not
    fail a later, valid call."""

    def __init__(self):
        self.msgs = []

    def append(self, m):
        self.msgs.append((time.monotonic(), m))

    def clear(self):
        self.msgs.clear()

    def __bool__(self):
        now = time.monotonic()
        self.msgs = [e for e in self.msgs if now - e[0] < 2.0]
        return bool(self.msgs)

    def __getitem__(self, i):
        return self.msgs[i][1]


_NOT_ON_GPU = _Pending()


def _raise_if_not_on_gpu() -> None:
    if _NOT_ON_GPU:
        msg = _NOT_ON_GPU[0]
        _NOT_ON_GPU.clear()
        raise HipLibraryError(msg)


def _dev(t: torch.Tensor, dtype: torch.dtype, name: str, numel: Optional[int] = None) -> int:
    if t is None:
        raise HipLibraryError(f"{name}: expected a tensor, got None")
    return tptr(t, name, dtype=dtype, numel=numel)


def on_device(dev):
    """`torch.cuda.device(dev)` for the launch; a no-op for a non-GPU device, so that the argument checks (tptr) are what raises
    when a CPU tensor reaches a wrapper."""
    import contextlib
    dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
    return torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()


def _stream() -> int:
    _raise_if_not_on_gpu()
    return torch.cuda.current_stream().cuda_stream


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------
def vq_prepare(codebook: torch.Tensor) -> torch.Tensor:
    """Build the kernel-side image of a codebook (K, C); reuse it until the codebook changes."""
    L = lib()
    k, c = codebook.shape
    wp = _dev(codebook, torch.float32, "codebook")
    nbytes = L.vqseg_vq_prepared_bytes(c, k)
    blob = _workspace(nbytes, codebook.device)
    with on_device(codebook.device):
        rc = L.vqseg_vq_prepare_f32(wp, c, k, blob.data_ptr(), nbytes, _stream())
    _check(rc, "vqseg_vq_prepare_f32")
    return blob


def vq_forward(rows: torch.Tensor, codebook: torch.Tensor, training: bool, commitment_weight: float,
               want_dmin: bool = False, prepared: Optional[torch.Tensor] = None):
    """rows (N, C) f32, codebook (K, C) f32 -> quant (N, C), idx (N,) i64, loss (1,), dead_pct (), [dmin (N,)]."""
    L = lib()
    n, c = rows.shape
    k = codebook.shape[0]
    bf16 = rows.dtype == torch.bfloat16                      # bf16 activations: same fp32 arithmetic, bf16 quant
    xp, wp = tptr(rows, "rows", bf=int(bf16), numel=n * c), _dev(codebook, torch.float32, "codebook", k * c)
    dev = rows.device
    quant = torch.empty_like(rows)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    scal = torch.empty(2, dtype=torch.float32, device=dev)
    dmin = torch.empty(n, dtype=torch.float32, device=dev) if want_dmin else None
    nbytes = L.vqseg_vq_workspace_bytes(n, c, k)
    ws = _workspace(nbytes, dev)
    with on_device(dev):
        fwd = L.vqseg_vq_forward_bf16 if bf16 else L.vqseg_vq_forward_f32
        rc = fwd(xp, wp, tptr(prepared, "prepared codebook", dtype=torch.uint8, numel=L.vqseg_vq_prepared_bytes(c, k)), n, c, k,
                 int(bool(training)), float(commitment_weight), quant.data_ptr(),
                 idx.data_ptr(), scal.data_ptr(), scal.data_ptr() + 4,
                 dmin.data_ptr() if want_dmin else None, ws.data_ptr(), nbytes, _stream())
    _check(rc, "vqseg_vq_forward_f32")
    out = (quant, idx, scal[0:1], scal[1])
    return out + (dmin,) if want_dmin else out


def bind(path: str) -> ctypes.CDLL:
    """another build of the same ABI (the per-workgroup timeline build) as a SECOND handle next to lib(): measurement code only"""
    handle = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = res, args
    return handle


def vq_forward_group(rows_list, codebooks, prepared_list, training: bool, commitment_weights, handle=None):
    """vq_forward for several independent layers with ONE distance + argmin launch (vqseg_vq_forward_group).
    -> list of (quant, idx, loss (1,), dead_pct ()) per level; bit-identical to per-level vq_forward calls.
    `handle`: another build of the library (bind()), for measurement code."""
    import numpy as np
    L = handle or lib()
    nl = len(rows_list)
    bf16 = rows_list[0].dtype == torch.bfloat16
    dev = rows_list[0].device
    outs, keep = [], []
    ptr = lambda ts: (c_void_p * nl)(*[t.data_ptr() for t in ts])
    ns = np.array([r.shape[0] for r in rows_list], dtype=np.int64)
    cs = np.array([r.shape[1] for r in rows_list], dtype=np.int32)
    ks = np.array([w.shape[0] for w in codebooks], dtype=np.int32)
    cw = np.array([float(w) for w in commitment_weights], dtype=np.float32)
    quants, idxs, scals, wss = [], [], [], []
    for r, w in zip(rows_list, codebooks):
        if r.dtype != rows_list[0].dtype:
            raise HipLibraryError("vq_forward_group: one row type per call")
        tptr(r, "rows", bf=int(bf16)), _dev(w, torch.float32, "codebook", w.shape[0] * r.shape[1])
        quants.append(torch.empty_like(r))
        idxs.append(torch.empty(r.shape[0], dtype=torch.int64, device=dev))
        scals.append(torch.empty(2, dtype=torch.float32, device=dev))
        wss.append(_workspace(L.vqseg_vq_workspace_bytes(r.shape[0], r.shape[1], w.shape[0]), dev))
    wsb = (c_size_t * nl)(*[w.numel() for w in wss])
    loss_p = (c_void_p * nl)(*[s.data_ptr() for s in scals])
    dead_p = (c_void_p * nl)(*[s.data_ptr() + 4 for s in scals])
    for w, p_ in zip(codebooks, prepared_list):
        tptr(p_, "prepared codebook", dtype=torch.uint8, numel=L.vqseg_vq_prepared_bytes(w.shape[1], w.shape[0]))
    with on_device(dev):
        rc = L.vqseg_vq_forward_group(nl, int(bf16), ptr(rows_list), ptr(codebooks), ptr(prepared_list), ns.ctypes.data, cs.ctypes.data,
                                      ks.ctypes.data, int(bool(training)), cw.ctypes.data, ptr(quants), ptr(idxs), loss_p, dead_p,
                                      ptr(wss), ctypes.cast(wsb, c_void_p), _stream())
    _check(rc, "vqseg_vq_forward_group")
    if FILTER_DIAG is not None and bf16:                     # diagnostics (tools/vq_filter_diag.py): per level (rows, channels, codes, counter of candidate pairs)
        for r, w, ws_ in zip(rows_list, codebooks, wss):
            off = L.vqseg_vq_filter_counter_offset(r.shape[0], r.shape[1], w.shape[0])
            FILTER_DIAG.append((r.shape[0], r.shape[1], w.shape[0], ws_[off:off + 256].view(torch.int32) if off else None))
    return [(q, i, s[0:1], s[1]) for q, i, s in zip(quants, idxs, scals)]


def set_option(key: str, value: int) -> int:
    """vqseg_set_option: returns the previous value"""
    prev = lib().vqseg_set_option(key.encode(), int(value))
    if prev < 0:
        raise HipLibraryError(f"vqseg_set_option: unknown option {key!r} or bad value {value}")
    return prev


def vq_assign(rows: torch.Tensor, codebook: torch.Tensor, want_dmin: bool = False,
              prepared: Optional[torch.Tensor] = None, want_filter_count: bool = False):
    L = lib()
    n, c = rows.shape
    k = codebook.shape[0]
    bf16 = rows.dtype == torch.bfloat16                      # bf16 activations: same fp32 arithmetic, bf16 quant
    xp, wp = tptr(rows, "rows", bf=int(bf16), numel=n * c), _dev(codebook, torch.float32, "codebook", k * c)
    dev = rows.device
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    dmin = torch.empty(n, dtype=torch.float32, device=dev) if want_dmin else None
    nbytes = L.vqseg_vq_workspace_bytes(n, c, k)
    ws = _workspace(nbytes, dev)
    with on_device(dev):
        fn = L.vqseg_vq_assign_bf16 if bf16 else L.vqseg_vq_assign_f32        # the entry point must match the row type
        rc = fn(xp, wp, tptr(prepared, "prepared codebook", dtype=torch.uint8, numel=L.vqseg_vq_prepared_bytes(c, k)), n, c, k,
                idx.data_ptr(), dmin.data_ptr() if want_dmin else None, ws.data_ptr(), nbytes, _stream())
    _check(rc, "vqseg_vq_assign_bf16" if bf16 else "vqseg_vq_assign_f32")
    if want_filter_count:                                    # diagnostics: candidate pairs the bf16 filter handed to the exact re-score (None: no filter)
        off = L.vqseg_vq_filter_counter_offset(n, c, k) if bf16 else 0
        amb = ws[off:off + 256].view(torch.int32).sum() if off else None      # 64 sub-list counters
        return (idx, dmin, amb) if want_dmin else (idx, amb)
    return (idx, dmin) if want_dmin else idx


def vq_backward(grad_quant: torch.Tensor, grad_loss: Optional[torch.Tensor], rows: torch.Tensor, quant: torch.Tensor,
                commitment_weight: float) -> torch.Tensor:
    L = lib()
    n, c = rows.shape
    gq = _dev(grad_quant, torch.float32, "grad_quant", n * c)
    gl = _dev(grad_loss, torch.float32, "grad_loss", 1) if grad_loss is not None else None
    gx = torch.empty_like(rows)
    with on_device(rows.device):
        rc = L.vqseg_vq_backward_f32(gq, gl, _dev(rows, torch.float32, "rows", n * c), _dev(quant, torch.float32, "quant", n * c),
                                     n, c, float(commitment_weight), gx.data_ptr(), _stream())
    _check(rc, "vqseg_vq_backward_f32")
    return gx


def vq_backward_bf16(grad_quant: torch.Tensor, grad_loss: Optional[torch.Tensor], rows: torch.Tensor, idx: torch.Tensor,
                     codebook: torch.Tensor, commitment_weight: float) -> torch.Tensor:
    """bf16 activations: grad_x = grad_quant + (2 w grad_loss / (N C)) (x - codebook[idx]), e re-read in fp32."""
    L = lib()
    n, c = rows.shape
    gq = _dev(grad_quant, torch.bfloat16, "grad_quant", n * c)
    gl = _dev(grad_loss, torch.float32, "grad_loss", 1) if grad_loss is not None else None
    gx = torch.empty_like(rows)
    if codebook.dim() != 2 or codebook.shape[1] != c:
        raise HipLibraryError(f"codebook: expected (K, {c}), got {tuple(codebook.shape)}")
    with on_device(rows.device):
        rc = L.vqseg_vq_backward_bf16(gq, gl, _dev(rows, torch.bfloat16, "rows", n * c), _dev(idx, torch.int64, "idx", n),
                                      _dev(codebook, torch.float32, "codebook"), n, c, float(commitment_weight), gx.data_ptr(),
                                      _stream())
    _check(rc, "vqseg_vq_backward_bf16")
    return gx


def kmeans(samples: torch.Tensor, means: torch.Tensor, iters: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Lloyd iterations in place on `means` (K, C); returns (means, bins (K,) i64)."""
    L = lib()
    n, c = samples.shape
    k = means.shape[0]
    sp, mp = _dev(samples, torch.float32, "samples", n * c), _dev(means, torch.float32, "means", k * c)
    bins = torch.zeros(k, dtype=torch.int64, device=samples.device)
    nbytes = L.vqseg_kmeans_workspace_bytes(n, c, k)
    ws = _workspace(nbytes, samples.device)
    with on_device(samples.device):
        rc = L.vqseg_kmeans_f32(sp, mp, bins.data_ptr(), n, c, k, int(iters), ws.data_ptr(), nbytes, _stream())
    _check(rc, "vqseg_kmeans_f32")
    return means, bins


def kmeans_accumulate(samples: torch.Tensor, means: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """This rank's per-cluster sums (K, C) f32 and counts (K,) i64 for one Lloyd iteration."""
    L = lib()
    n, c = samples.shape
    k = means.shape[0]
    sp, mp = _dev(samples, torch.float32, "samples", n * c), _dev(means, torch.float32, "means", k * c)
    sums = torch.empty(k, c, dtype=torch.float32, device=samples.device)
    counts = torch.empty(k, dtype=torch.int64, device=samples.device)
    nbytes = L.vqseg_kmeans_workspace_bytes(n, c, k)
    ws = _workspace(nbytes, samples.device)
    with on_device(samples.device):
        rc = L.vqseg_kmeans_accumulate_f32(sp, mp, n, c, k, sums.data_ptr(), counts.data_ptr(), ws.data_ptr(), nbytes,
                                           _stream())
    _check(rc, "vqseg_kmeans_accumulate_f32")
    return sums, counts


def kmeans_finalize(sums: torch.Tensor, counts: torch.Tensor, means: torch.Tensor) -> torch.Tensor:
    L = lib()
    k, c = means.shape
    with on_device(means.device):
        rc = L.vqseg_kmeans_finalize_f32(_dev(sums, torch.float32, "sums", k * c), _dev(counts, torch.int64, "counts", k),
                                         _dev(means, torch.float32, "means", k * c), c, k, _stream())
    _check(rc, "vqseg_kmeans_finalize_f32")
    return means


def vq_code_sums(rows: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-code sums (K, C) f32 and counts (K,) i64 of rows (N, C) f32 / bf16 under the assignment idx (N,) i64."""
    L = lib()
    n, c = rows.shape
    if rows.dtype not in (torch.float32, torch.bfloat16):
        raise HipLibraryError(f"rows: expected torch.float32 or torch.bfloat16, got {rows.dtype}")
    ip = _dev(idx, torch.int64, "idx", n)
    sums = torch.empty(k, c, dtype=torch.float32, device=rows.device)
    counts = torch.empty(k, dtype=torch.int64, device=rows.device)
    nbytes = L.vqseg_kmeans_workspace_bytes(n, c, k)
    ws = _workspace(nbytes, rows.device)
    with on_device(rows.device):
        rc = L.vqseg_vq_code_sums(int(rows.dtype == torch.bfloat16), tptr(rows, "rows", bf=int(rows.dtype == torch.bfloat16), numel=n * c), ip, n, c, k, sums.data_ptr(), counts.data_ptr(),
                                  ws.data_ptr(), nbytes, _stream())
    _check(rc, "vqseg_vq_code_sums")
    return sums, counts


def vq_ema_update(cluster_size: torch.Tensor, embed_avg: torch.Tensor, codebook: torch.Tensor, sums: torch.Tensor,
                  counts: torch.Tensor, decay: float, eps: float) -> None:
    """In place: moving counts / sums and the codebook they imply (include/vqseg.h: vqseg_vq_ema_update_f32)."""
    L = lib()
    k, c = codebook.shape
    if cluster_size.shape != (k,) or embed_avg.shape != (k, c) or sums.shape != (k, c) or counts.shape != (k,):
        raise ValueError("vq_ema_update: shapes must be (K,), (K, C), (K, C), (K, C), (K,)")
    scratch = torch.empty(1, dtype=torch.float32, device=codebook.device)
    with on_device(codebook.device):
        rc = L.vqseg_vq_ema_update_f32(_dev(cluster_size, torch.float32, "cluster_size"), _dev(embed_avg, torch.float32, "embed_avg"),
                                       _dev(codebook, torch.float32, "codebook"), _dev(sums, torch.float32, "sums"),
                                       _dev(counts, torch.int64, "counts"), c, k, float(decay), float(eps), scratch.data_ptr(),
                                       _stream())
    _check(rc, "vqseg_vq_ema_update_f32")


def profile_begin(capacity: int = 4096) -> None:
    _check(lib().vqseg_profile_begin(int(capacity)), "vqseg_profile_begin")


def conv_profile_begin(capacity: int = 65536) -> None:
    _check(lib().vqseg_conv_profile_begin(int(capacity)), "vqseg_conv_profile_begin")


def conv_profile_collect(capacity: int = 65536, with_shape: bool = False):
    """-> list of (algorithmic flops, kind = KH * 100 + {0 bf16, 1 precise, 2 split-3}, milliseconds) per convolution launch"""
    import numpy as np
    fl = np.zeros(capacity, dtype=np.float64)
    kd = np.zeros(capacity, dtype=np.int32)
    ms = np.zeros(capacity, dtype=np.float32)
    sh = np.zeros((capacity, 4), dtype=np.int32)
    cnt = lib().vqseg_conv_profile_collect(capacity, fl.ctypes.data, kd.ctypes.data, ms.ctypes.data, sh.ctypes.data if with_shape else None)
    if cnt < 0:
        _check(cnt, "vqseg_conv_profile_collect")
    if with_shape:
        return [(float(fl[i]), int(kd[i]), float(ms[i]), tuple(int(v) for v in sh[i])) for i in range(cnt)]
    return [(float(fl[i]), int(kd[i]), float(ms[i])) for i in range(cnt)]


def profile_collect(capacity: int = 4096, with_kind: bool = False):
    """-> list of (n_rows, channels, n_codes, milliseconds[, kind]) for every assign launch since profile_begin
    (kind 0: exact kernel on f32 rows, 1: exact kernel on bf16 rows, 2: bf16 candidate filter + exact re-score)."""
    import numpy as np
    n = np.zeros(capacity, dtype=np.int64)
    c = np.zeros(capacity, dtype=np.int32)
    k = np.zeros(capacity, dtype=np.int32)
    ms = np.zeros(capacity, dtype=np.float32)
    kd = np.zeros(capacity, dtype=np.int32)
    cnt = lib().vqseg_profile_collect(capacity, n.ctypes.data, c.ctypes.data, k.ctypes.data, ms.ctypes.data, kd.ctypes.data)
    if cnt < 0:
        _check(cnt, "vqseg_profile_collect")
    if with_kind:
        return [(int(n[i]), int(c[i]), int(k[i]), float(ms[i]), int(kd[i])) for i in range(cnt)]
    return [(int(n[i]), int(c[i]), int(k[i]), float(ms[i])) for i in range(cnt)]
