"""ctypes wrapper of oracle/vq_chain.c (CPU oracle, TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libvq_chain.so")
_lib = None

ORDER_NATURAL, ORDER_MFMA8 = 0, 1


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _DIR, "-s"])
        _lib = ctypes.CDLL(_SO)
        _lib.vq_chain_assign.restype = ctypes.c_int
        _lib.vq_chain_assign.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def assign(rows, codebook, order: int = ORDER_NATURAL):
    """rows (N, C) f32, codebook (K, C) f32 (numpy or CPU torch) -> idx (N,) int64, dmin (N,) f32 (numpy)."""
    x = np.ascontiguousarray(np.asarray(rows, dtype=np.float32))
    w = np.ascontiguousarray(np.asarray(codebook, dtype=np.float32))
    n, c = x.shape
    k = w.shape[0]
    idx = np.empty(n, dtype=np.int64)
    dmin = np.empty(n, dtype=np.float32)
    rc = _load().vq_chain_assign(x.ctypes.data, w.ctypes.data, n, c, k, order, idx.ctypes.data, dmin.ctypes.data)
    if rc != 0:
        raise ValueError("vq_chain_assign: bad arguments")
    return idx, dmin
