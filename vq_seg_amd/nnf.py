"""Functional building blocks of the encoder/decoder, executed by the gfx950 HIP kernels behind
include/vqseg.h (conv_kernels.hip, nn_kernels.hip).  torch only owns memory and the autograd tape.

Layout: activations are NHWC in memory ("channels_last" views of logically NCHW tensors).
Precision follows the activation dtype: float32 -> "precise" kernels (bf16x3 split MFMA, fp32 storage;
the parity mode), bfloat16 -> "fast" kernels (bf16 storage and operands, fp32 accumulate).

Every function here has exactly one implementation; CPU tensors are refused (no fallback).
"""
from __future__ import annotations

import contextlib
import ctypes
from typing import Optional
import weakref

import torch

from . import _hip
from ._hip import _check, _dev, _stream, lib
from ._wcache import cache_of as _cache_of, invalidate as invalidate_weight_caches  # noqa: F401


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def _rows(x: torch.Tensor) -> torch.Tensor:
    """logical (N, C, H, W) -> contiguous (N, H, W, C) view (free for channels_last tensors)."""
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _nchw(rows: torch.Tensor) -> torch.Tensor:
    return rows.permute(0, 3, 1, 2)


def _is_bf16(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 0
    raise _hip.HipLibraryError(f"activations must be float32 (precise mode) or bfloat16 (fast mode), got {t.dtype}")


_T = _hip.tptr                # every tensor handed to the library goes through this (dtype / density / element count / device checks)


def _w16(t, name, numel=None, at_least=False):
    """packed bf16 weight images travel as int16 tensors"""
    return _T(t, name, dtype=torch.int16, numel=numel, at_least=at_least)


def _f32(t, name, numel=None):
    return _T(t, name, dtype=torch.float32, numel=numel)


def _strided_f32(t, name, numel):
    """(B, C, H, W) fp32 logits addressed through the explicit (batch, class, pixel) strides passed along (NCHW and NHWC storage both
    qualify): a valid torch view bounds every such address by construction, so only type, element count and device are checked."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
        raise _hip.HipLibraryError(f"{name}: expected a float32 tensor, got {getattr(t, 'dtype', type(t).__name__)}")
    if t.numel() != numel:
        raise _hip.HipLibraryError(f"{name}: the sizes passed along need {numel} elements, the tensor has {t.numel()}")
    if not t.is_cuda:                                        # deferred like _hip.tptr's: the call's other tensors are checked first
        _hip._NOT_ON_GPU.append(f"{name}: the HIP path needs a tensor on a 'cuda' (ROCm) device; got {t.device}. There is no CPU fallback.")
        return 0
    return t.data_ptr()


def act_dtype() -> torch.dtype:
    """bf16 'fast' mode under torch.autocast (the reference trainer's `autocast(enabled=half)`,
    train_vqreptunet1x1v2.py:151 -- fp16 there, bf16 here: the ROCm-native half type), else fp32."""
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return torch.float32


def _pack_all(weight: torch.Tensor, kind):
    """bf16-mode images of a k x k (k = 1 or 3) conv weight -- kind "fwd", "tr" (data gradient) or ("s3", c1) -- built together
    in ONE launch (vqseg_conv_pack_all_f32) with every other kind this weight has been asked for before (the set survives the
    invalidation after each optimiser step, so from the second step on a layer costs one pack launch per step instead of three).
    None: not that kind of weight (the caller packs the single image)."""
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3] or weight.shape[2] not in (1, 3) or not py_opt("py_pack_all", 1):
        return None
    kinds = getattr(weight, "_vq_kinds", None)
    if kinds is None:
        kinds = weight._vq_kinds = set()
    if kind not in kinds:
        if isinstance(kind, tuple) and any(isinstance(k, tuple) and k != kind for k in kinds):
            return None                                      # a second concat split for the same weight: single-image path
        kinds.add(kind)
    cache = _cache_of(weight)
    imgs = cache.get("all")
    if imgs is None or kind not in imgs:
        w = weight.detach()
        w = w if w.is_contiguous() else w.contiguous()
        cout, cin, k, _ = w.shape
        cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
        s3k = next((kk for kk in kinds if isinstance(kk, tuple)), None)
        if s3k is not None and (cin % 32 or s3k[1] % 32):
            return None
        imgs = {}
        if "fwd" in kinds:
            imgs["fwd"] = torch.empty(cout * k * k * cin_p, dtype=torch.int16, device=w.device)
        if "tr" in kinds:
            imgs["tr"] = torch.empty(cin * k * k * cout_p, dtype=torch.int16, device=w.device)
        if s3k is not None:
            imgs[s3k] = torch.empty(cout * k * k * 3 * cin, dtype=torch.int16, device=w.device)
        with _hip.on_device(w.device):
            _check(lib().vqseg_conv_pack_all_f32(_f32(w, "weight", cout * cin * k * k), cout, cin, k, s3k[1] if s3k is not None else cin,
                                                 _w16(imgs.get("fwd"), "fwd image", cout * k * k * cin_p),
                                                 _w16(imgs.get("tr"), "transposed image", cin * k * k * cout_p),
                                                 _w16(imgs.get(s3k), "split-3 image", cout * k * k * 3 * cin) if s3k is not None else None,
                                                 _stream()), "vqseg_conv_pack_all_f32")
        cache["all"] = imgs
    return imgs[kind]


def packed_weights(weight: torch.Tensor, precise: bool, transpose_flip: bool):
    """MFMA-side image of an nn.Conv2d weight, rebuilt only when the parameter changes (_wcache: version counter, storage,
    and every optimiser step)."""
    if not precise:
        img = _pack_all(weight, "tr" if transpose_flip else "fwd")
        if img is not None:
            return img, None
    cache = _cache_of(weight)
    k = (precise, transpose_flip)
    if k not in cache:
        w = weight.detach()
        w = w if w.is_contiguous() else w.contiguous()
        cout, cin, kh, kw = w.shape
        n = lib().vqseg_conv_packed_elems(cout, cin, kh, kw, int(transpose_flip))
        hi = torch.empty(n, dtype=torch.int16, device=w.device)
        lo = torch.empty(n, dtype=torch.int16, device=w.device) if precise else None
        with _hip.on_device(w.device):
            _check(lib().vqseg_conv_pack_weights_f32(_f32(w, "weight", cout * cin * kh * kw), cout, cin, kh, kw, int(transpose_flip),
                                                     _w16(hi, "packed hi", n), _w16(lo, "packed lo", n), _stream()), "vqseg_conv_pack_weights_f32")
        cache[k] = (hi, lo)
    return cache[k]


def _conv_raw(x_rows, x2_rows, c1, w_hi, w_lo, out_shape, stat, n, h, w, cin, cout, kh, kw, stride, pad, reflect, up,
              ho, wo, w_offset_elems=0):
    precise = x_rows.dtype == torch.float32
    y = torch.empty(out_shape, dtype=x_rows.dtype, device=x_rows.device)
    esz = 2
    bfa = 0 if precise else 1
    wneed = w_offset_elems + cout * kh * kw * ((cin + 31) // 32 * 32)        # the [Cout][KH][KW][Cin ^ 32] slice the kernel reads
    with _hip.on_device(x_rows.device):
        rc = lib().vqseg_conv2d_f(_T(x_rows, "conv input", bf=bfa, numel=n * h * w * c1), _T(x2_rows, "conv input 2", bf=bfa, numel=n * h * w * (cin - c1)),
                                  c1, _w16(w_hi, "packed weights", wneed, at_least=True) + w_offset_elems * esz,
                                  (_w16(w_lo, "packed weights (lo)", wneed, at_least=True) + w_offset_elems * esz) if w_lo is not None else None,
                                  _T(y, "conv output", bf=bfa, numel=n * ho * wo * cout),
                                  _f32(stat, "BN partials", lib().vqseg_conv_stat_slots(n * ho * wo, cout) * 2 * cout if stat is not None else None),
                                  n, h, w, cin, cout, kh, kw, stride, pad, int(reflect), up, ho, wo, int(precise), _stream())
    _check(rc, "vqseg_conv2d_f")
    return y


def _bn_sync(bn, backward: bool):
    """The BatchNorm module's own hand-over counters for the fused merge + finalize / reduce + finalize launches (include/vqseg.h,
    `sync`): zeros between launches, never shared by concurrent launches -- one buffer per module, one half per direction (a
    module's forwards and backwards are ordered on its network's stream).  OFF by default: measured on the bench step (r3, same
    box, alternating) the one-launch forms are 0.3-2 % SLOWER than the two-launch ones -- the last workgroup's fold is a serial
    tail read through device-coherent loads, and the launches it saves were not on the critical path (the other network's stream
    covers them).  VQSEG_OPTS=py_bn_fused=1 switches them on."""
    if not py_opt("py_bn_fused", 0):
        return None
    c = bn.num_features
    g = lib().vqseg_bn_sync_ints(c)
    buf = getattr(bn, "_vq_sync", None)
    dev = bn.weight.device
    if buf is None or buf.device != dev or buf.numel() != 2 * g:
        buf = bn._vq_sync = torch.zeros(2 * g, dtype=torch.int32, device=dev)
    return buf[g:] if backward else buf[:g]


def _check_fused_bn(rc: int, what: str, bn) -> None:
    """`_check` for the two entry points that take a module's hand-over counters: a launch that failed midway may leave them
    non-zero, and every later fused launch on that module would then never elect a last workgroup (scale / shift / dgamma / dbeta
    silently stale).  On a non-zero return code the module's counters are zeroed before the error is raised."""
    if rc != 0:
        buf = getattr(bn, "_vq_sync", None)
        if buf is not None:
            buf.zero_()
    _check(rc, what)


# Side streams for the weight-gradient kernels: {cuda_stream handle of a network's stream: torch.cuda.Stream}.  A trainer that owns
# the parameters' .grad storage registers one per network stream (trainer.CPSTrainer), adds it to the buckets' producer streams
# and joins it before the optimiser step.  Empty: everything stays on the calling stream.
WGRAD_SIDE_STREAMS: dict = {}


def _wgrad_side_stream(dev):
    if not WGRAD_SIDE_STREAMS:
        return None
    return WGRAD_SIDE_STREAMS.get(torch.cuda.current_stream(dev).cuda_stream)


def _out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


# ------------------------------------------------------------------------------------------------
# Gradient sinks.  A trainer that owns the parameters' .grad storage (trainer.GradBuckets: flat fp32 buckets) marks a
# parameter with `p._vq_grad_sink = callback-or-None`; the weight-gradient kernels then ADD into p.grad themselves
# (`accumulate` flag of the C entry points) and autograd gets None for that parameter -- one tiny add kernel per
# parameter and backward pass less.  The callback (if any) is told, as a post-accumulate-grad hook would be.
# ------------------------------------------------------------------------------------------------
def py_opt(key: str, default: int) -> int:
    """host-side A/B switch from VQSEG_OPTS ("py_<name>=<int>", see _hip.lib)"""
    lib()
    return _hip.PY_OPTS.get(key, default)


# Parameters with a queued first use of a two-use weight gradient (see _ConvBNAct.backward); flush_pending_wgrads() launches
# whatever is still queued (a forward whose output did not reach the loss) -- trainers call it after backward().
_PENDING_WGRADS: set = set()


def _wgrad_launch(geom, a, b, gw, accumulate, stream):
    """vqseg_conv2d_wgrad2_f over use `a` = (g_y, x, x2, n) [and use `b`] into gw; `stream`: the stream a queued use belongs to."""
    h, w, c1, cin, ho, wo, cout, kh, kw, stride, pad, reflect, precise, patches_of = geom
    g_y, xr, x2r, n = a
    gb, xb, x2b, nb = b if b is not None else (None, None, None, 0)
    L = lib()
    dev = g_y.device
    bf = 0 if precise else 1
    m = n * ho * wo
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        if patches_of:
            okh, okw, ocin = patches_of[0], patches_of[1], patches_of[2]
            wkh = wkw = 1
            im2col, cin_out, st_, pd_, rf_ = 1, ocin, 1, 0, 0
            gnum = cout * ocin * okh * okw
        else:
            okh, okw, wkh, wkw = kh, kw, kh, kw
            im2col, cin_out, st_, pd_, rf_ = 0, cin, stride, pad, reflect
            gnum = cout * cin * kh * kw
        nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(n + nb, h, w, cin, ho, wo, cout, wkh, wkw)
        wsw = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _hip.on_device(dev):
            _check(L.vqseg_conv2d_wgrad2_f(_T(g_y, "conv output gradient", bf=bf, numel=m * cout), _T(xr, "conv input", bf=bf, numel=n * h * w * c1),
                                           _T(x2r, "conv input 2", bf=bf, numel=n * h * w * (cin - c1)), n,
                                           _T(gb, "conv output gradient (second use)", bf=bf, numel=nb * ho * wo * cout),
                                           _T(xb, "conv input (second use)", bf=bf, numel=nb * h * w * c1),
                                           _T(x2b, "conv input 2 (second use)", bf=bf, numel=nb * h * w * (cin - c1)), nb,
                                           c1, h, w, cin, ho, wo, cout, okh, okw, st_, pd_, rf_, int(precise), cin_out, im2col, int(accumulate),
                                           _T(wsw, "wgrad workspace", dtype=torch.uint8, numel=nbytes), nbytes,
                                           _f32(gw, "weight gradient", gnum), _stream()), "vqseg_conv2d_wgrad2_f")


def flush_pending_wgrads() -> int:
    """Launch every queued first use that never met its second (returns how many).  After this no Parameter holds activations."""
    n = 0
    for p in list(_PENDING_WGRADS):
        pend = getattr(p, "_vq_wgrad_pending", None)
        p._vq_wgrad_pending = None
        if pend is not None:
            _wgrad_launch(pend[0], pend[1], None, p.grad, True, pend[2])
            n += 1
    _PENDING_WGRADS.clear()
    return n


def drop_pending_wgrads() -> None:
    """forget queued uses without launching them (start of a step: whatever an aborted backward left behind)"""
    for p in list(_PENDING_WGRADS):
        p._vq_wgrad_pending = None
    _PENDING_WGRADS.clear()


class GradLink:
    """Carries the gradient of a block's identity branch from the backward of its LAST conv (`link_out`: where the
    residual add happened) to the backward of its FIRST conv (`link_in`), whose data-gradient kernel adds it in its
    epilogue -- instead of autograd materialising both gradients of the block input and adding them in a separate
    pass.  Valid when both convolutions read the same tensor (identity shortcut).
    Projection shortcut: the shortcut's convolution sends the gradient of ITS input (`link_x`) the same way.  Its backward
    has no dependency on the first conv's, so the order is the autograd engine's choice: a producer that arrives after the
    consumer has run (`closed`) simply returns its gradient to autograd."""
    __slots__ = ("g", "closed", "leftover", "bits", "takes_bits", "__weakref__")

    def __init__(self):
        self.g = None
        self.closed = False
        self.leftover = None
        self.bits = None            # r4: `g` is the block output's UNMASKED gradient, `bits` the ReLU mask of the block's last BatchNorm
        self.takes_bits = False     # set by the first conv's forward: its data-gradient epilogue can apply `bits` itself


def _mask_with_bits(g, bits):
    """g .* bits (bit i % 8 of byte i / 8 over the flat index) with stock operators -- the fallback of a bit-field shortcut gradient
    whose consumer turned out not to fuse it"""
    sh = torch.arange(8, device=g.device, dtype=torch.uint8)
    keep = torch.bitwise_and(torch.bitwise_right_shift(bits.view(-1, 1), sh), 1).view(g.shape).to(torch.bool)
    return torch.where(keep, g, torch.zeros((), dtype=g.dtype, device=g.device))


# ------------------------------------------------------------------------------------------------
# Fan-in fusion (r4).  A tensor with TWO consumers (an encoder feature feeds the next encoder stage AND the decoder / its VQ layer;
# the decoder output feeds the head AND the prototype loss) gets two gradients that autograd adds in a separate pass over the
# tensor (5 such adds per backward pass, ~2 ms per training step).  With fusion on, the producer of the tensor tags it with a
# GradLink (`t._vq_fanin`); the consumer whose backward runs FIRST deposits its gradient there (and returns nothing to autograd),
# the one that runs LAST adds it inside its own kernel: the stride-2 1x1 projection's data gradient accumulates in place
# (vqseg_conv2d_dgrad_s2_f accumulate), the max-pool and head backward kernels take an addend.  Results are bit-identical to
# autograd's add.  Order is a data dependency for the encoder features (the decoder's backward is upstream of the encoder's) and
# the engine's priority order for the decoder output; a depositor that comes after the absorber (`closed`) returns its gradient
# to autograd as usual.  A deposit that nobody absorbs would be a lost gradient: only callers that check for it after backward
# (trainer.CPSTrainer: check_fanin_consumed) switch the fusion on.
# ------------------------------------------------------------------------------------------------
_FANIN_ON = False
_FANIN_OPEN: set = set()                                  # links holding a deposited, not yet absorbed gradient


def set_fanin_fusion(on: bool) -> bool:
    global _FANIN_ON
    prev = _FANIN_ON
    _FANIN_ON = bool(on) and py_opt("py_fanin", 1) == 1
    return prev


def fanin_tag(t):
    """producer side: mark `t` (a tensor with two consumers) for fan-in fusion in the coming backward"""
    if _FANIN_ON and torch.is_tensor(t) and t.requires_grad and torch.is_grad_enabled():
        t._vq_fanin = GradLink()
    return t


def _fanin_of(t):
    return getattr(t, "_vq_fanin", None) if (_FANIN_ON and torch.is_tensor(t)) else None


def _fanin_deposit(link, g_rows) -> bool:
    """first consumer: hand the gradient (NHWC rows) to the link; False: the absorber has already run"""
    if link is None or link.closed or link.g is not None:
        return False
    link.g = g_rows
    _FANIN_OPEN.add(link)
    return True


def _fanin_take(link, shape, dtype):
    """last consumer: the deposited gradient if it fits (shape, dtype, dense), else None; either way the link is closed.
    A deposit that does not fit is returned through `link.leftover` for a plain add."""
    if link is None:
        return None
    link.closed = True
    g, link.g = link.g, None
    _FANIN_OPEN.discard(link)
    if g is None:
        return None
    if tuple(g.shape) == tuple(shape) and g.dtype == dtype and g.is_contiguous():
        return g
    link.leftover = g
    return None


def check_fanin_consumed() -> None:
    """after backward: every deposited gradient must have been absorbed (else it never reached its tensor)"""
    if _FANIN_OPEN:
        n = len(_FANIN_OPEN)
        for l_ in list(_FANIN_OPEN):
            l_.g = None
        _FANIN_OPEN.clear()
        raise RuntimeError(f"fan-in fusion: {n} deposited gradient(s) were never absorbed -- a consumer's backward did not run; "
                           f"switch the fusion off (VQSEG_OPTS=py_fanin=0) for graphs that use only part of the model's outputs")


_UNIT_AFFINE = {}


def _unit_affine(dev, c):
    key = (str(dev), c)
    if key not in _UNIT_AFFINE:
        _UNIT_AFFINE[key] = (torch.ones(c, dtype=torch.float32, device=dev), torch.zeros(c, dtype=torch.float32, device=dev))
        torch.cuda.current_stream(dev).synchronize()       # once per (device, width): other streams may read them next
    return _UNIT_AFFINE[key]


def _sink_ready(p) -> bool:
    g = getattr(p, "grad", None)
    return hasattr(p, "_vq_grad_sink") and g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.is_cuda


def _sink_use(*params) -> None:
    """forward (called where the grad mode is still visible): one more gradient contribution will arrive for each
    of these parameters in the coming backward"""
    if torch.is_grad_enabled():
        for p in params:
            if p.requires_grad and hasattr(p, "_vq_grad_sink"):
                p._vq_uses = getattr(p, "_vq_uses", 0) + 1


def _sink_done(p) -> None:
    """backward: one contribution has been added into p.grad; the callback fires with the last one"""
    p._vq_uses = getattr(p, "_vq_uses", 1) - 1
    cb = p._vq_grad_sink
    if cb is not None and p._vq_uses <= 0:
        cb(p)


# ------------------------------------------------------------------------------------------------
# "Split-3" activations: the fp32-precision no-grad eval forward on the bf16 kernels (include/vqseg.h, "Split-3").
# The reference trainers run their two pseudo-label forwards in fp32, OUTSIDE autocast (train_vqreptunet1x1v2.py:143-149).
# The "precise" kernels do that with on-the-fly bf16 hi/lo splits inside a register-staged kernel; here the SAME three products
# run on the LDS-DMA / patch-reuse bf16 kernels by keeping every activation of such a forward as [hi | lo] bf16 (the
# convolution's K loop runs over [hi | lo | hi], reading hi twice, against weights [w_hi | w_hi | w_lo]).
# An `S3` object stands in for the logical (N, C, H, W) fp32 tensor between the layers of ONE model forward; it never leaves
# the model (the model's `encode` opens the scope, the 1x1 head / the VQ layers merge back to fp32).
# ------------------------------------------------------------------------------------------------
class S3:
    """rows: (N, H, W, 2C) bf16 contiguous = [hi | lo] of the logical fp32 tensor (N, C, H, W)."""
    __slots__ = ("rows", "c")

    def __init__(self, rows: torch.Tensor, c: int):
        self.rows, self.c = rows, c

    @property
    def shape(self):
        n, h, w, _ = self.rows.shape
        return torch.Size((n, self.c, h, w))

    dtype = torch.float32                                    # what the tensor logically is
    is_cuda = True
    requires_grad = False

    @property
    def device(self):
        return self.rows.device

    def float(self) -> torch.Tensor:
        """merge back: logical (N, C, H, W) fp32 tensor (channels_last in memory)"""
        n, h, w, _ = self.rows.shape
        out = torch.empty((n, h, w, self.c), dtype=torch.float32, device=self.rows.device)
        with _hip.on_device(out.device):
            _check(lib().vqseg_s3_merge_f(_T(self.rows, "split-3 rows", bf=2, numel=n * h * w * 2 * self.c), n * h * w, self.c,
                                          _f32(out, "merged rows", n * h * w * self.c), _stream()), "vqseg_s3_merge_f")
        return _nchw(out)


_S3_SCOPE = False


class s3_scope:
    """Inside this scope a no-grad, eval-mode, fp32 (not autocast) stem starts a split-3 forward (opened by the VQ-UNet's own
    `encode`; a bare `model.encoder(x)` call therefore keeps returning plain fp32 tensors).  VQSEG_OPTS=py_s3_eval=0 disables."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled) and py_opt("py_s3_eval", 1) == 1

    def __enter__(self):
        global _S3_SCOPE
        self.prev = _S3_SCOPE
        _S3_SCOPE = self.enabled
        return self

    def __exit__(self, *exc):
        global _S3_SCOPE
        _S3_SCOPE = self.prev
        return False


def to_s3(x) -> "S3":
    """logical (N, C, H, W) fp32 (or bf16) tensor -> split-3"""
    if isinstance(x, S3):
        return x
    xr = _rows(x.float())
    n, h, w, c = xr.shape
    out = torch.empty((n, h, w, 2 * c), dtype=torch.bfloat16, device=xr.device)
    with _hip.on_device(xr.device):
        _check(lib().vqseg_s3_split_f(_f32(xr, "rows", n * h * w * c), n * h * w, c, _T(out, "split-3 rows", bf=2, numel=n * h * w * 2 * c), _stream()),
               "vqseg_s3_split_f")
    return S3(out, c)


def from_s3(x):
    return x.float() if isinstance(x, S3) else x


def _s3_weights(weight: torch.Tensor, c1: int, as_1x1_cols: int = 0) -> torch.Tensor:
    """[w_hi | w_hi | w_lo] image per concat segment (vqseg_conv_pack_weights_s3_f32), cached on the Parameter."""
    if not as_1x1_cols:
        img = _pack_all(weight, ("s3", c1))
        if img is not None:
            return img
    cache = _cache_of(weight)
    k = ("s3", c1, as_1x1_cols)
    if k not in cache:
        w = weight.detach()
        if as_1x1_cols:                                      # the stem: [64][3][7][7] as a 1x1 convolution over kp patch columns
            cout, cin, kh, kw = w.shape
            w2 = torch.zeros(cout, as_1x1_cols, 1, 1, dtype=torch.float32, device=w.device)
            w2[:, :kh * kw * cin, 0, 0] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
            w = w2
        w = w if w.is_contiguous() else w.contiguous()
        cout, cin, kh, kw = w.shape
        out = torch.empty(cout * kh * kw * 3 * cin, dtype=torch.int16, device=w.device)
        with _hip.on_device(w.device):
            _check(lib().vqseg_conv_pack_weights_s3_f32(_f32(w, "weight", cout * cin * kh * kw), cout, cin, c1, kh, kw,
                                                        _w16(out, "split-3 image", cout * kh * kw * 3 * cin), _stream()),
                   "vqseg_conv_pack_weights_s3_f32")
        cache[k] = out
    return cache[k]


def _conv_bn_act_s3(x: "S3", x2, residual, conv, bn, relu, kernel_1x1_cols: int = 0) -> "S3":
    """eval-mode Conv -> BN -> [+ residual] -> [ReLU] on split-3 tensors: the convolution's fused epilogue (precise = 2)."""
    n, h, w, c3 = x.rows.shape
    c1 = x.c
    x2s = to_s3(x2) if x2 is not None else None
    cin = c1 + (x2s.c if x2s is not None else 0)
    cout = conv.weight.shape[0]
    if kernel_1x1_cols:
        kh = kw = 1
        stride, pad, reflect = 1, 0, False
    else:
        kh, kw = conv.weight.shape[2], conv.weight.shape[3]
        stride, pad = conv.stride[0], conv.padding[0]
        reflect = conv.padding_mode == "reflect" and pad > 0
    ho, wo = _out_size(h, kh, stride, pad), _out_size(w, kw, stride, pad)
    dev = x.rows.device
    L = lib()
    wimg = _s3_weights(conv.weight, c1, kernel_1x1_cols)
    res = to_s3(residual) if residual is not None else None
    coef = torch.empty(4, cout, dtype=torch.float32, device=dev)
    out = torch.empty((n, ho, wo, 2 * cout), dtype=torch.bfloat16, device=dev)
    with _hip.on_device(dev):
        _check(L.vqseg_bn_finalize_f(None, n * ho * wo, cout, _f32(bn.weight, "bn.weight", cout), _f32(bn.bias, "bn.bias", cout),
                                     _f32(bn.running_mean, "bn.running_mean", cout), _f32(bn.running_var, "bn.running_var", cout),
                                     float(bn.momentum), float(bn.eps), 0, _f32(coef[0], "scale", cout),
                                     _f32(coef[1], "shift", cout), _f32(coef[2], "mean", cout), _f32(coef[3], "invstd", cout), None, None, _stream()),
               "vqseg_bn_finalize_f")
        _check(L.vqseg_conv2d_affine_f(_T(x.rows, "split-3 input", bf=2, numel=n * h * w * 2 * c1),
                                       _T(x2s.rows, "split-3 input 2", bf=2, numel=n * h * w * 2 * (cin - c1)) if x2s is not None else None, c1,
                                       _w16(wimg, "split-3 image", cout * kh * kw * 3 * cin), None,
                                       _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout),
                                       _T(res.rows, "split-3 residual", bf=2, numel=n * ho * wo * 2 * cout) if res is not None else None, int(relu),
                                       _T(out, "split-3 output", bf=2, numel=n * ho * wo * 2 * cout), n, h, w, cin, cout, kh, kw, stride, pad,
                                       int(reflect), ho, wo, 2, _stream()),
               "vqseg_conv2d_affine_f (split-3)")
    return S3(out, cout)


# ------------------------------------------------------------------------------------------------
# Conv (no bias) -> BatchNorm -> [+ residual] -> [ReLU], optional channel concat of two inputs
# ------------------------------------------------------------------------------------------------
class _ConvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, residual, weight, gamma, beta, bn, stride, pad, reflect, relu, patches_of, fuse_eval=False,
                link_in=None, link_out=None, link_x=None, fan_x=None, fan_x2=None):
        """x (N,C1,H,W) [+ x2 (N,C2,H,W)] -> out (N,Cout,Ho,Wo).  `patches_of` = (kh, kw, cin, stride, pad, reflect,
        H, W) when x is an im2col patch matrix of the stem (then the convolution itself is 1x1)."""
        xr = _rows(x)
        x2r = _rows(x2) if x2 is not None else None
        bf = _is_bf16(xr)
        precise = not bf
        n, h, w, c1 = xr.shape
        # r4, the stem without its patch matrix (patches_of[8] = (ho, wo, kp)): x is the fp32 IMAGE; the convolution gathers its operands
        # from it (vqseg_stem7_conv_f, bit-identical to the 1x1 convolution over the patch matrix), the activations are bf16; backward
        # builds the patch matrix for the weight gradient then (shared between the two networks as before)
        stem_img = bool(patches_of) and len(patches_of) > 8
        stem_x = None
        if stem_img:
            stem_x, _STEM_X_ARG[0] = _STEM_X_ARG[0], None
            h, w, c1 = patches_of[8]
            bf, precise = True, False
        adt = torch.bfloat16 if stem_img else xr.dtype      # activation dtype of y / out
        cin = c1 + (x2r.shape[3] if x2r is not None else 0)
        cout = weight.shape[0]
        kh, kw = (1, 1) if patches_of else (weight.shape[2], weight.shape[3])
        ho, wo = _out_size(h, kh, stride, pad), _out_size(w, kw, stride, pad)
        m = n * ho * wo
        training = bool(bn.training)
        L = lib()
        dev = xr.device
        if stem_img:
            w_hi, w_lo = _stem_weights_fused(weight, False), None
        elif patches_of:
            w_hi, w_lo = _stem_weights(weight, precise, cin)
        else:
            w_hi, w_lo = packed_weights(weight, precise, False)
        coef = torch.empty(4, cout, dtype=torch.float32, device=dev)       # scale, shift, mean, invstd
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not used by the path")
        if fuse_eval and not training:
            # eval-mode BatchNorm is a fixed per-channel affine map: it rides in the convolution's epilogue together with
            # the residual add and the ReLU (the no-grad pseudo-label passes; nothing is kept for a backward)
            rr = _rows(residual) if residual is not None else None
            if rr is not None and rr.dtype != xr.dtype:
                raise _hip.HipLibraryError("residual dtype differs from the activation dtype")
            out = torch.empty((n, ho, wo, cout), dtype=adt, device=dev)
            wneed = cout * kh * kw * ((cin + 31) // 32 * 32)
            with _hip.on_device(dev):
                _check(L.vqseg_bn_finalize_f(None, m, cout, _f32(gamma, "bn.weight", cout), _f32(beta, "bn.bias", cout),
                                             _f32(bn.running_mean, "bn.running_mean", cout), _f32(bn.running_var, "bn.running_var", cout),
                                             float(bn.momentum), float(bn.eps), 0, _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout),
                                             _f32(coef[2], "mean", cout), _f32(coef[3], "invstd", cout), None, None, _stream()), "vqseg_bn_finalize_f")
                if stem_img:
                    _check(L.vqseg_stem7_conv_f(0, _f32(xr, "image"), _w16(w_hi, "stem weight image", cout * 176), _T(out, "conv output", bf=1, numel=m * cout), None,
                                                _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout), int(relu), n, patches_of[6], patches_of[7],
                                                int(patches_of[5]), _stream()), "vqseg_stem7_conv_f")
                    return _nchw(out)
                _check(L.vqseg_conv2d_affine_f(_T(xr, "conv input", bf=bf, numel=n * h * w * c1), _T(x2r, "conv input 2", bf=bf, numel=n * h * w * (cin - c1)),
                                               c1, _w16(w_hi, "packed weights", wneed), _w16(w_lo, "packed weights (lo)", wneed),
                                               _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout),
                                               _T(rr, "residual", bf=bf, numel=m * cout), int(relu), _T(out, "conv output", bf=bf, numel=m * cout),
                                               n, h, w, cin, cout, kh, kw, stride, pad, int(reflect), ho, wo, int(precise), _stream()),
                       "vqseg_conv2d_affine_f")
            return _nchw(out)
        stat = torch.empty(L.vqseg_conv_stat_slots(m, cout) * 2 * cout, dtype=torch.float32, device=dev) if training else None
        if stem_img:
            y = torch.empty((n, ho, wo, cout), dtype=adt, device=dev)
            with _hip.on_device(dev):
                _check(L.vqseg_stem7_conv_f(0, _f32(xr, "image"), _w16(w_hi, "stem weight image", cout * 176),
                                            _T(y, "conv output", bf=1, numel=m * cout), _f32(stat, "BN partials"), None, None, 0, n, patches_of[6],
                                            patches_of[7], int(patches_of[5]), _stream()), "vqseg_stem7_conv_f")
        else:
            y = _conv_raw(xr, x2r, c1, w_hi, w_lo, (n, ho, wo, cout), stat, n, h, w, cin, cout, kh, kw, stride, pad, reflect, 1,
                          ho, wo)
        with _hip.on_device(dev):
            _check_fused_bn(L.vqseg_bn_finalize_f(_f32(stat, "BN partials"), m, cout, _f32(gamma, "bn.weight", cout), _f32(beta, "bn.bias", cout),
                                         _f32(bn.running_mean, "bn.running_mean", cout), _f32(bn.running_var, "bn.running_var", cout),
                                         float(bn.momentum), float(bn.eps), int(training), _f32(coef[0], "scale", cout),
                                         _f32(coef[1], "shift", cout), _f32(coef[2], "mean", cout), _f32(coef[3], "invstd", cout),
                                         _T(bn.num_batches_tracked, "bn.num_batches_tracked", dtype=torch.int64, numel=1) if training else None,
                                         _T(_bn_sync(bn, False), "bn sync", dtype=torch.int32) if training else None,
                                         _stream()),   # += 1 in the kernel
                   "vqseg_bn_finalize_f", bn)
            rr = _rows(residual) if residual is not None else None
            if rr is not None and rr.dtype != y.dtype:
                raise _hip.HipLibraryError("residual dtype differs from the activation dtype")
            out = torch.empty_like(y)
            # residual layers in bf16: the ReLU mask goes to the backward as a bit field (1/16 of `out`'s bytes), see vqseg.h
            bits = (torch.empty(m * cout // 8, dtype=torch.uint8, device=dev)
                    if (rr is not None and relu and bf and cout % 8 == 0 and training and any(ctx.needs_input_grad)
                        and py_opt("py_bn_bits", 1)) else None)
            if bits is not None:
                _check(L.vqseg_bn_apply_bits_f(_T(y, "conv output", bf=1, numel=m * cout), _T(rr, "residual", bf=1, numel=m * cout),
                                               _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout), m, cout,
                                               _T(out, "BN output", bf=1, numel=m * cout),
                                               _T(bits, "ReLU mask bits", dtype=torch.uint8, numel=m * cout // 8), _stream()),
                       "vqseg_bn_apply_bits_f")
            else:
                _check(L.vqseg_bn_apply_f(bf, _T(y, "conv output", bf=bf, numel=m * cout), _T(rr, "residual", bf=bf, numel=m * cout),
                                          _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout), m, cout, int(relu),
                                          _T(out, "BN output", bf=bf, numel=m * cout), _stream()), "vqseg_bn_apply_f")
        ctx.mask_bits = bits is not None
        if link_in is not None:                                             # see backward: the zero-padded stride-1 data gradient + shortcut
            link_in.takes_bits = bool(bf and x2r is None and stride == 1 and not reflect and cin % 8 == 0 and py_opt("py_gres_bits", 1))
        ctx.save_for_backward(xr, x2r, y, bits if bits is not None else out, coef, weight, gamma)
        ctx.stem_x = stem_x if stem_img else None                           # the image tensor OBJECT the patch matrix is shared under
        ctx.params = (weight, gamma, beta)                                  # the Parameter objects (see grad sinks)
        ctx.bn = bn
        ctx.links = (link_in, link_out, link_x)
        ctx.fans = (fan_x, fan_x2)                                          # fan-in links: absorb into x's gradient / deposit x2's
        ctx.cfg = (stride, pad, bool(reflect), bool(relu), training, residual is not None, patches_of, (n, h, w, c1, cin, cout,
                                                                                                         kh, kw, ho, wo))
        return _nchw(out)

    @staticmethod
    def backward(ctx, g_out):
        xr, x2r, y, out, coef, weight, gamma = ctx.saved_tensors
        stride, pad, reflect, relu, training, has_res, patches_of, (n, h, w, c1, cin, cout, kh, kw, ho, wo) = ctx.cfg
        L = lib()
        dev = y.device
        if patches_of and len(patches_of) > 8:                              # forward ran from the image: the weight gradient's patch matrix now
            img = xr
            okh, okw, ocin, os_, op_, orf, ih_, iw_ = patches_of[:8]
            xr = _shared_stem_patches(ctx.stem_x if ctx.stem_x is not None else img, (torch.bfloat16, okh, okw, os_, op_, orf),
                                      lambda: _stem_patches(img, torch.bfloat16, okh, okw, os_, op_, orf, ho, wo, cin))
        bf = _is_bf16(y)
        precise = not bf
        m = n * ho * wo
        g = _rows(g_out)
        if g.dtype != y.dtype:
            g = g.to(y.dtype)
        g_y = torch.empty_like(y)
        link_in, link_out, link_x = ctx.links
        # r4: the residual branch's gradient g_out .* mask is not stored when the block's first conv will apply the mask bits itself
        bits_to_link = bool(has_res and ctx.mask_bits and link_out is not None and link_out.takes_bits and not link_out.closed)
        g_res = torch.empty_like(y) if (has_res and not bits_to_link) else None
        ws = torch.empty(L.vqseg_bn_backward_workspace_floats(m, cout), dtype=torch.float32, device=dev)
        p_w, p_g, p_b = ctx.params
        sink_bn = _sink_ready(p_g) and _sink_ready(p_b) and ctx.needs_input_grad[4] and ctx.needs_input_grad[5]
        sink_w = _sink_ready(p_w) and ctx.needs_input_grad[3]
        dgb = None if sink_bn else torch.empty(2, cout, dtype=torch.float32, device=dev)
        dgamma, dbeta = (p_g.grad, p_b.grad) if sink_bn else (dgb[0], dgb[1])
        with _hip.on_device(dev):
            # without a residual the ReLU mask is recomputed from y with the forward's scale / shift: `out` is not re-read
            if ctx.mask_bits:                                               # `out` holds the forward's mask bits here
                _check_fused_bn(L.vqseg_bn_backward_bits_f(
                    _T(g, "output gradient", bf=1, numel=m * cout), _T(out, "ReLU mask bits", dtype=torch.uint8, numel=m * cout // 8),
                    _T(y, "conv output", bf=1, numel=m * cout), _f32(coef[2], "mean", cout), _f32(coef[3], "invstd", cout),
                    _f32(gamma.detach(), "bn.weight", cout), m, cout, int(training), int(sink_bn), _f32(ws, "BN backward workspace"),
                    _f32(dgamma, "bn.weight.grad", cout), _f32(dbeta, "bn.bias.grad", cout),
                    _T(g_y, "conv output gradient", bf=1, numel=m * cout), _T(g_res, "residual gradient", bf=1, numel=m * cout),
                    _T(_bn_sync(ctx.bn, True), "bn sync", dtype=torch.int32), _stream()), "vqseg_bn_backward_bits_f", ctx.bn)
            else:
                _check_fused_bn(L.vqseg_bn_backward_f(bf, _T(g, "output gradient", bf=bf, numel=m * cout),
                                           _T(out, "BN output", bf=bf, numel=m * cout) if has_res else None, _T(y, "conv output", bf=bf, numel=m * cout),
                                           _f32(coef[2], "mean", cout), _f32(coef[3], "invstd", cout), _f32(gamma.detach(), "bn.weight", cout),
                                           _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout), m, cout, int(relu), int(training), int(sink_bn),
                                           _f32(ws, "BN backward workspace"), _f32(dgamma, "bn.weight.grad", cout), _f32(dbeta, "bn.bias.grad", cout),
                                           _T(g_y, "conv output gradient", bf=bf, numel=m * cout), _T(g_res, "residual gradient", bf=bf, numel=m * cout),
                                           _T(_bn_sync(ctx.bn, True), "bn sync", dtype=torch.int32), _stream()), "vqseg_bn_backward_f", ctx.bn)
        if sink_bn:
            _sink_done(p_g), _sink_done(p_b)
        if link_out is not None and has_res:
            link_out.g = g if bits_to_link else g_res                       # picked up by the block's first conv (GradLink)
            link_out.bits = out if bits_to_link else None
        # ---- weight gradient.  With a grad sink (the result is ADDED into the trainer's bucket, nothing returns to autograd) and a
        # registered side stream, the weight-gradient kernel and its slab sum leave the network's stream: nothing downstream in
        # backward depends on them, and as filler work they cover the latency-bound links of the main chain (BatchNorm statistics
        # folds, finalizes, packs: skipping those ~950 launches -- wrong results, timing only -- was worth 11 ms of a 163 ms step)
        gw = p_w.grad if sink_w else torch.empty(weight.shape, dtype=torch.float32, device=dev)
        wside = _wgrad_side_stream(dev) if sink_w else None
        if wside is not None:
            main_s = torch.cuda.current_stream(dev)
            wside.wait_event(main_s.record_event())                             # g_y (bn_backward above) is complete
            for t_ in (g_y, xr, x2r):
                if t_ is not None:
                    t_.record_stream(wside)                                     # allocator: not reusable before the side stream is done
        geom = (h, w, c1, cin, ho, wo, cout, kh, kw, stride, pad, int(reflect), int(precise), patches_of)
        use = (g_y, xr, x2r, n)
        # ---- two-use weights (r4): a weight used by two training forwards of the step (the labelled and the unlabelled batch of a
        # CPS iteration) gets ONE launch over both uses -- the first use is queued on the Parameter, the second runs
        # vqseg_conv2d_wgrad2_f over the virtual batch: twice the contraction length per workgroup, half the slab sums
        pend = getattr(p_w, "_vq_wgrad_pending", None) if sink_w else None
        if pend is not None:
            p_w._vq_wgrad_pending = None
            _PENDING_WGRADS.discard(p_w)
        if pend is not None and (pend[0] != geom or pend[2] != torch.cuda.current_stream(dev) or wside is not None):
            _wgrad_launch(pend[0], pend[1], None, p_w.grad, True, pend[2])      # not pairable after all: the queued use on its own
            pend = None
        if sink_w and pend is None and wside is None and getattr(p_w, "_vq_uses", 1) >= 2 and py_opt("py_wgrad_pair", 1):
            p_w._vq_wgrad_pending = (geom, use, torch.cuda.current_stream(dev))
            _PENDING_WGRADS.add(p_w)
        else:
            with (torch.cuda.stream(wside) if wside is not None else contextlib.nullcontext()):
                _wgrad_launch(geom, pend[1] if pend is not None else use, use if pend is not None else None, gw, sink_w, None)
        if sink_w:
            _sink_done(p_w)
        # ---- data gradient(s): the same implicit-GEMM kernel on g_y with tap-flipped, transposed weights
        gx = gx2 = None
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if (need1 or need2) and not patches_of:
            t_hi, t_lo = packed_weights(weight, precise, True)                 # [Cin][kh][kw][Cout]
            taps = kh * kw
            hp, wp = (h + 2 * pad, w + 2 * pad) if reflect else (h, w)         # reflect: gradient of the padded input first
            dpad = (kh - 1) if reflect else (kh - 1 - pad)

            extra = link_in.g if link_in is not None else None          # residual-branch gradient of the same block input
            ebits = link_in.bits if link_in is not None else None
            if link_in is not None:
                link_in.g = link_in.bits = None
                link_in.closed = True
            if ebits is not None and not (not reflect and stride == 1 and x2r is None and bf and extra.shape == (n, hp, wp, c1)):
                extra, ebits = _mask_with_bits(extra, ebits), None       # not the fusing path below after all

            def dgrad(c_lo, c_cnt):
                nonlocal extra
                # reflect padding, 3x3 / stride 1 (every Bottleneck conv2): the padded gradient's INTERIOR is the zero-padded data
                # gradient (fast patch kernel, unpadded grid); only its border ring is evaluated as a full correlation and folded
                # onto rows 1 / H-2 and columns 1 / W-2 (vqseg_reflect_ring_f) -- no (H+2) x (W+2) tensor, no fold pass over it
                ring = (reflect and stride == 1 and kh == 3 and kw == 3 and pad == 1 and bf and x2r is None and cout % 64 == 0
                        and c_cnt % 8 == 0 and h >= 4 and w >= 4 and py_opt("py_reflect_ring", 1) == 1)
                if ring:
                    gp = None
                    if extra is not None and extra.shape == (n, h, w, c_cnt):
                        one, zero = _unit_affine(dev, c_cnt)
                        gp = torch.empty((n, h, w, c_cnt), dtype=g_y.dtype, device=dev)
                        with _hip.on_device(dev):
                            tneed = c_cnt * 9 * ((cout + 31) // 32 * 32)
                            _check(L.vqseg_conv2d_affine_f(_T(g_y, "conv output gradient", bf=bf, numel=m * cout), None, cout,
                                                           _w16(t_hi, "transposed image", tneed), None, _f32(one, "unit scale", c_cnt),
                                                           _f32(zero, "zero shift", c_cnt), _T(extra, "shortcut gradient", bf=bf, numel=n * h * w * c_cnt), 0,
                                                           _T(gp, "input gradient", bf=bf, numel=n * h * w * c_cnt), n, ho, wo, cout, c_cnt,
                                                           3, 3, 1, 1, 0, h, w, 0, _stream()), "vqseg_conv2d_affine_f")
                        extra = None
                    else:
                        gp = _conv_raw(g_y, None, cout, t_hi, t_lo, (n, h, w, c_cnt), None, n, ho, wo, cout, c_cnt, 3, 3, 1, 1, False, 1, h, w)
                    rbuf = torch.empty((n, 2 * (w + 2) + 2 * h, c_cnt), dtype=g_y.dtype, device=dev)
                    with _hip.on_device(dev):
                        _check(L.vqseg_reflect_ring_f(_T(g_y, "conv output gradient", bf=1, numel=m * cout),
                                                      _w16(t_hi, "transposed image", c_cnt * 9 * ((cout + 31) // 32 * 32)),
                                                      _T(rbuf, "gradient ring", bf=1, numel=rbuf.numel()), _T(gp, "input gradient", bf=1, numel=n * h * w * c_cnt),
                                                      n, h, w, cout, c_cnt, _stream()), "vqseg_reflect_ring_f")
                    return gp
                if extra is not None and not reflect and stride == 1 and x2r is None and extra.shape == (n, hp, wp, c_cnt):
                    # the data gradient and the residual-branch gradient meet in the convolution's epilogue (one add pass less)
                    one, zero = _unit_affine(dev, c_cnt)
                    gp = torch.empty((n, hp, wp, c_cnt), dtype=g_y.dtype, device=dev)
                    with _hip.on_device(dev):
                        tneed = c_cnt * kh * kw * ((cout + 31) // 32 * 32)
                        if ebits is not None:
                            _check(L.vqseg_conv2d_affine_bits_f(
                                _T(g_y, "conv output gradient", bf=1, numel=m * cout), _w16(t_hi, "transposed image", tneed),
                                _f32(one, "unit scale", c_cnt), _f32(zero, "zero shift", c_cnt),
                                _T(extra, "block output gradient", bf=1, numel=n * hp * wp * c_cnt),
                                _T(ebits, "ReLU mask bits", dtype=torch.uint8, numel=n * hp * wp * c_cnt // 8),
                                _T(gp, "input gradient", bf=1, numel=n * hp * wp * c_cnt), n, ho, wo, cout, c_cnt, kh, kw, dpad, hp, wp,
                                _stream()), "vqseg_conv2d_affine_bits_f")
                        else:
                            _check(L.vqseg_conv2d_affine_f(_T(g_y, "conv output gradient", bf=bf, numel=m * cout), None, cout,
                                                         _w16(t_hi, "transposed image", tneed), _w16(t_lo, "transposed image (lo)", tneed),
                                                         _f32(one, "unit scale", c_cnt), _f32(zero, "zero shift", c_cnt),
                                                         _T(extra, "shortcut gradient", bf=bf, numel=n * hp * wp * c_cnt), 0,
                                                         _T(gp, "input gradient", bf=bf, numel=n * hp * wp * c_cnt), n, ho, wo, cout, c_cnt,
                                                         kh, kw, 1, dpad, 0, hp, wp, int(precise), _stream()), "vqseg_conv2d_affine_f")
                    extra = None
                    return gp
                if (stride == 2 and x2r is None and kh == 3 and kw == 3 and pad == 1 and bf and h == 2 * ho and w == 2 * wo and h >= 4 and w >= 4
                        and cout % 64 == 0 and c_cnt % 8 == 0 and c_cnt >= 64 and py_opt("py_dgrad_s2", 1) and py_opt("py_dgrad_s2_fold", 1)):
                    # r4: the four parity classes in ONE launch, written straight into the unpadded gradient (reflect padding: the
                    # padded top row / left column go to a small ring behind the pixel rows and are added onto row 1 / column 1) --
                    # no (H+2) x (W+2) tensor, no fold / crop pass over it
                    s_hi, _s_lo = _s2_weights(weight, precise)
                    rows = int(L.vqseg_conv2d_dgrad_s2_fold_rows(n, h, w, int(reflect)))
                    buf = torch.empty((rows, c_cnt), dtype=g_y.dtype, device=dev)
                    with _hip.on_device(dev):
                        _check(L.vqseg_conv2d_dgrad_s2_fold_f(_T(g_y, "conv output gradient", bf=1, numel=m * cout),
                                                              _w16(s_hi, "parity-class image", L.vqseg_conv_packed_s2_elems(cout, c_cnt, 3)),
                                                              _T(buf, "input gradient (+ ring)", bf=1, numel=rows * c_cnt),
                                                              n, ho, wo, cout, c_cnt, h, w, int(reflect), _stream()), "vqseg_conv2d_dgrad_s2_fold_f")
                    return buf[:n * h * w].view(n, h, w, c_cnt)
                if stride == 2 and x2r is None and kh == kw and ((kh == 1 and pad == 0) or (kh == 3 and pad == 1)) and py_opt("py_dgrad_s2", 1):
                    # stride-2 layer: parity classes of the output pixel instead of a dilated gradient grid (vqseg_conv2d_dgrad_s2_f)
                    s_hi, s_lo = _s2_weights(weight, precise)
                    gh, gw_ = (h + 2, w + 2) if kh == 3 else (h, w)          # k = 3: the padded input's grid
                    # fan-in (1x1 projection of a stage's first block): the input's OTHER consumer (decoder skip / VQ layer) has
                    # deposited its gradient; this data gradient accumulates into it in place -- no memset, no add pass
                    acc_g = _fanin_take(fan_x, (n, gh, gw_, c_cnt), g_y.dtype) if (kh == 1 and c_cnt <= 4096) else None
                    gp = acc_g if acc_g is not None else torch.empty((n, gh, gw_, c_cnt), dtype=g_y.dtype, device=dev)
                    with _hip.on_device(dev):
                        sneed = L.vqseg_conv_packed_s2_elems(cout, c_cnt, kh)
                        _check(L.vqseg_conv2d_dgrad_s2_f(_T(g_y, "conv output gradient", bf=bf, numel=m * cout), _w16(s_hi, "parity-class image", sneed),
                                                         _w16(s_lo, "parity-class image (lo)", sneed), _T(gp, "input gradient", bf=bf, numel=n * gh * gw_ * c_cnt),
                                                         n, ho, wo, cout, c_cnt, kh, gh, gw_, int(precise), int(acc_g is not None), _stream()),
                               "vqseg_conv2d_dgrad_s2_f")
                    if kh == 3 and not reflect:                              # zero padding: the gradient of the padded border is dropped
                        return gp[:, 1:h + 1, 1:w + 1, :].contiguous()
                else:
                    gp = _conv_raw(g_y, None, cout, t_hi, t_lo, (n, hp, wp, c_cnt), None, n, ho, wo, cout, c_cnt, kh, kw, 1, dpad, False,
                                   stride, hp, wp, w_offset_elems=c_lo * taps * ((cout + 31) // 32 * 32))
                if not reflect:
                    return gp
                if pad != 1:
                    raise NotImplementedError("reflect-padding data gradient is implemented for pad == 1")
                gxx = torch.empty((n, h, w, c_cnt), dtype=gp.dtype, device=dev)
                with _hip.on_device(dev):
                    _check(L.vqseg_reflect_fold_f(bf, _T(gp, "padded input gradient", bf=bf, numel=n * (h + 2) * (w + 2) * c_cnt), n, h, w, c_cnt,
                                                  _T(gxx, "input gradient", bf=bf, numel=n * h * w * c_cnt), _stream()), "vqseg_reflect_fold_f")
                return gxx

            fan_x, fan_x2 = ctx.fans
            if need1:
                g1 = dgrad(0, c1)
                if extra is not None:                                       # link not fusable here: plain add
                    g1 = g1 + extra
                    extra = None
                if fan_x is not None:                                       # a deposit the kernel path above could not absorb
                    rest = _fanin_take(fan_x, g1.shape, g1.dtype)
                    rest = rest if rest is not None else fan_x.leftover
                    fan_x.leftover = None
                    if rest is not None:
                        g1 = g1 + rest.reshape(g1.shape).to(g1.dtype)
                if link_x is not None and not link_x.closed and x2r is None:
                    link_x.g = g1                                           # the block's first conv adds it (GradLink)
                else:
                    gx = _nchw(g1)
            if need2 and x2r is not None:
                g2 = dgrad(c1, cin - c1)
                if g2.dtype == x2r.dtype and _fanin_deposit(fan_x2, g2):    # x2's other consumer adds it inside its own kernel
                    gx2 = None
                else:
                    gx2 = _nchw(g2)
        g_res_out = _nchw(g_res) if (has_res and link_out is None) else None
        return (gx, gx2, g_res_out, None if sink_w else gw, None if sink_bn else dgb[0],
                None if sink_bn else dgb[1], None, None, None, None, None, None, None, None, None, None, None, None)


def _s2_weights(weight, precise):
    """parity-class sub-images of a stride-2 layer's weight for its data gradient (vqseg_conv_pack_weights_s2_f32)"""
    cache = _cache_of(weight)
    k = ("s2", precise)
    if k not in cache:
        w = weight.detach()
        w = w if w.is_contiguous() else w.contiguous()
        cout, cin, kh, _kw = w.shape
        nel = lib().vqseg_conv_packed_s2_elems(cout, cin, kh)
        hi = torch.empty(nel, dtype=torch.int16, device=w.device)
        lo = torch.empty(nel, dtype=torch.int16, device=w.device) if precise else None
        with _hip.on_device(w.device):
            _check(lib().vqseg_conv_pack_weights_s2_f32(_f32(w, "weight", cout * cin * kh * kh), cout, cin, kh, _w16(hi, "parity-class image", nel),
                                                        _w16(lo, "parity-class image (lo)", nel), _stream()),
                   "vqseg_conv_pack_weights_s2_f32")
        cache[k] = (hi, lo)
    return cache[k]


def _stem_weights(weight, precise, kp):
    """[64][3][7][7] -> packed [64][kp] rows ((kh, kw, ci) columns, zero padded to kp)."""
    cache = _cache_of(weight)
    k = ("stem", precise, kp)
    if k not in cache:
        # view the [Cout][Cin][KH][KW] stem weight as a 1x1 convolution over kp = pad32(KH*KW*Cin) patch columns
        cout, cin, kh, kw = weight.shape
        w2 = torch.zeros(cout, kp, 1, 1, dtype=torch.float32, device=weight.device)
        w2[:, :kh * kw * cin, 0, 0] = weight.detach().permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
        cache[k] = packed_weights(w2, precise, False)
        cache[("stem_keepalive", precise, kp)] = w2
    return cache[k]


def conv_bn_act(x, conv, bn, training=None, relu=True, residual=None, x2=None, link_in=None, link_out=None, link_x=None, absorb_fanin=False):
    """Conv2d (no bias; zero or reflect padding) -> BatchNorm2d -> [+ residual] -> [ReLU] on the HIP kernels.
    `x2`: second input whose channels follow x's (the decoder's concat).  `training` is ignored: the
    BatchNorm module's own mode decides (nn.BatchNorm2d semantics)."""
    if conv.bias is not None:
        raise NotImplementedError("conv_bn_act: the reference's fused blocks have no conv bias")
    if isinstance(x, S3) or isinstance(x2, S3) or isinstance(residual, S3):
        # the split-3 kernels take 32-channel K chunks per concat segment and 16-byte [hi | lo] output rows (vqseg_conv2d_affine_f,
        # precise == 2); any other width (e.g. decoder_channels ending in 16) merges back to fp32 and runs on the precise kernels,
        # which handle every Cin % 4 == 0 -- same values to 2^-16, just slower
        c1 = x.shape[1]
        c2 = x2.shape[1] if x2 is not None else 0
        s3_ok = c1 % 32 == 0 and c2 % 32 == 0 and conv.weight.shape[0] % 8 == 0
        if not bn.training and not torch.is_grad_enabled() and s3_ok:
            return _conv_bn_act_s3(to_s3(x), x2, residual, conv, bn, relu)
        x, x2, residual = from_s3(x), (from_s3(x2) if x2 is not None else None), (from_s3(residual) if residual is not None else None)
    if not x.is_cuda:
        raise _hip.HipLibraryError(f"the HIP path needs 'cuda' (ROCm) tensors, got {x.device}; there is no CPU fallback")
    pad = conv.padding[0]
    _sink_use(conv.weight, bn.weight, bn.bias)
    # fan-in links (see set_fanin_fusion): `absorb_fanin` -- this convolution's data gradient takes in the gradient x's other
    # consumer deposited; a tagged x2 (the decoder's skip input) gets ITS gradient deposited instead of returned to autograd
    return _ConvBNAct.apply(x, x2, residual, conv.weight, bn.weight, bn.bias, bn, conv.stride[0], pad,
                            conv.padding_mode == "reflect" and pad > 0, relu, None, not bn.training and not torch.is_grad_enabled(),
                            link_in, link_out, link_x, _fanin_of(x) if absorb_fanin else None, _fanin_of(x2))


# Within ONE training step the same image tensor goes through several stems: both networks of a CPS pair see the same
# batches, and the unlabelled batch passes twice (pseudo-label forward, training forward).  The patch matrix depends on the
# image alone, so a caller may open a sharing scope for the step (trainer.CPSTrainer does): forwards on the SAME tensor
# object (same version) inside the scope reuse one matrix.  Nothing is kept across scopes -- every step's images are
# unfolded in that step.
_STEM_SHARE = None


def stem_share_begin():
    global _STEM_SHARE
    _STEM_SHARE = []


def stem_share_end():
    global _STEM_SHARE
    _STEM_SHARE = None


def _shared_stem_patches(x, cfg, make):
    if _STEM_SHARE is None:
        return make()
    cur = torch.cuda.current_stream(x.device)
    for ref, ver, key, patches, ev, st in _STEM_SHARE:
        if ref() is x and ver == x._version and key == cfg:
            if st != cur:                                   # produced on the other network's stream
                cur.wait_event(ev)
                patches.record_stream(cur)
            return patches
    patches = make()
    ev = torch.cuda.Event()
    ev.record(cur)
    _STEM_SHARE.append((weakref.ref(x), x._version, cfg, patches, ev, cur))
    return patches


_STEM_X_ARG = [None]            # hands the image tensor OBJECT (patch-sharing identity) to _ConvBNAct.forward past autograd's argument handling


def _stem_patches(xr, dt, kh, kw, s, p, reflect, ho, wo, kp):
    """the stem's im2col patch matrix [n][ho][wo][kp] of the fp32 image rows xr"""
    n, h, w, cin = xr.shape
    out = torch.empty((n, ho, wo, kp), dtype=dt, device=xr.device)
    with _hip.on_device(xr.device):
        _check(lib().vqseg_im2col_f(int(dt == torch.bfloat16), _f32(xr, "image", n * h * w * cin), n, h, w, cin, kh, kw, s, p, int(reflect), ho, wo, kp,
                                    _T(out, "patches", bf=int(dt == torch.bfloat16), numel=n * ho * wo * kp), _stream()), "vqseg_im2col_f")
    return out


def _stem_weights_fused(weight, s3: bool):
    """vqseg_stem7_conv_f's weight image: [64][176] bf16, column kh * 24 + kw * 3 + ci (each kernel row's 21 taps padded to 24), zeros
    elsewhere; split-3: [64][2][176] = hi | lo.  Cached on the Parameter (_wcache: rebuilt after every optimiser step)."""
    cache = _cache_of(weight)
    k = ("stem_fused", bool(s3))
    if k not in cache:
        cout, cin, kh, kw = weight.shape
        w = weight.detach().float().permute(0, 2, 3, 1).reshape(cout, kh, kw * cin)           # [co][kh][(kw, ci)]
        wp = torch.zeros(cout, 176, dtype=torch.float32, device=weight.device)
        wp[:, :kh * 24].view(cout, kh, 24)[:, :, :kw * cin] = w
        hi = wp.to(torch.bfloat16)
        img = torch.cat([hi, (wp - hi.float()).to(torch.bfloat16)], dim=1) if s3 else hi
        cache[k] = img.contiguous().view(torch.int16)
    return cache[k]


def stem_conv_bn_act(x, conv, bn):
    """7x7 stride-2 stem on a 3-channel fp32 image: im2col patch matrix (zero / reflect padding) + 1x1 MFMA conv."""
    if not x.is_cuda:
        raise _hip.HipLibraryError(f"the HIP path needs 'cuda' (ROCm) tensors, got {x.device}; there is no CPU fallback")
    xr = _rows(x.float())
    n, h, w, cin = xr.shape
    kh, kw = conv.kernel_size
    s, p = conv.stride[0], conv.padding[0]
    reflect = conv.padding_mode == "reflect"
    ho, wo = _out_size(h, kh, s, p), _out_size(w, kw, s, p)
    kp = (kh * kw * cin + 31) // 32 * 32
    dt = act_dtype()
    # r4: the convolution straight from the image (vqseg_stem7_conv_f) where its tiling fits: 128 output pixels of one row per workgroup.
    # It removes 7.5 GB of patch-matrix traffic per step and its kernel takes 197 us against 318 us.  Measured +-0 in the step when it was
    # built (152.7 vs 152.5 ms) and -0.9 ms on the final r4 build (145.3 vs 146.1 ms, LEDGER r4): default from then on; the patch-matrix
    # path (VQSEG_OPTS=py_stem_fused=0) stays for output widths that are not multiples of 128 and for the weight gradient
    fused_ok = ((kh, kw, cin) == (7, 7, 3) and s == 2 and p == 3 and wo % 128 == 0 and h >= 4 and w >= 4 and conv.weight.shape[0] == 64
                and py_opt("py_stem_fused", 1) == 1)
    if _S3_SCOPE and dt == torch.float32 and not bn.training and not torch.is_grad_enabled() and (kh, kw, cin) == (7, 7, 3):
        # fp32-precision eval forward: split-3 patch rows (64-column multiple: the LDS-DMA kernels), then everything stays split-3
        kp3 = (kh * kw * cin + 63) // 64 * 64

        def make3():
            out = torch.empty((n, ho, wo, 2 * kp3), dtype=torch.bfloat16, device=x.device)
            with _hip.on_device(x.device):
                _check(lib().vqseg_im2col_f(2, _f32(xr, "image", n * h * w * cin), n, h, w, cin, kh, kw, s, p, int(reflect), ho, wo, kp3,
                                            _T(out, "split-3 patches", bf=2, numel=n * ho * wo * 2 * kp3), _stream()), "vqseg_im2col_f (split-3)")
            return out

        if fused_ok:                                             # r4: no split-3 patch matrix (1.6 GB written, read by both networks)
            cout = conv.weight.shape[0]
            wimg = _stem_weights_fused(conv.weight, True)
            coef = torch.empty(4, cout, dtype=torch.float32, device=x.device)
            out = torch.empty((n, ho, wo, 2 * cout), dtype=torch.bfloat16, device=x.device)
            with _hip.on_device(x.device):
                _check(lib().vqseg_bn_finalize_f(None, n * ho * wo, cout, _f32(bn.weight, "bn.weight", cout), _f32(bn.bias, "bn.bias", cout),
                                                 _f32(bn.running_mean, "bn.running_mean", cout), _f32(bn.running_var, "bn.running_var", cout),
                                                 float(bn.momentum), float(bn.eps), 0, _f32(coef[0], "scale", cout), _f32(coef[1], "shift", cout),
                                                 _f32(coef[2], "mean", cout), _f32(coef[3], "invstd", cout), None, None, _stream()), "vqseg_bn_finalize_f")
                _check(lib().vqseg_stem7_conv_f(1, _f32(xr, "image", n * h * w * cin), _w16(wimg, "stem weight image (hi | lo)", cout * 2 * 176),
                                                _T(out, "split-3 output", bf=2, numel=n * ho * wo * 2 * cout), None, _f32(coef[0], "scale", cout),
                                                _f32(coef[1], "shift", cout), 1, n, h, w, int(reflect), _stream()), "vqseg_stem7_conv_f (split-3)")
            return S3(out, cout)
        patches = _shared_stem_patches(x, ("s3", kh, kw, s, p, reflect), make3)
        return _conv_bn_act_s3(S3(patches, kp3), None, None, conv, bn, True, kernel_1x1_cols=kp3)

    if fused_ok and dt == torch.bfloat16 and kp == 160:
        _sink_use(conv.weight, bn.weight, bn.bias)
        _STEM_X_ARG[0] = x
        return _ConvBNAct.apply(_nchw(xr), None, None, conv.weight, bn.weight, bn.bias, bn, 1, 0, False, True,
                                (kh, kw, cin, s, p, reflect, h, w, (ho, wo, kp)), not bn.training and not torch.is_grad_enabled())
    patches = _shared_stem_patches(x, (dt, kh, kw, s, p, reflect), lambda: _stem_patches(xr, dt, kh, kw, s, p, reflect, ho, wo, kp))
    _sink_use(conv.weight, bn.weight, bn.bias)
    return _ConvBNAct.apply(_nchw(patches), None, None, conv.weight, bn.weight, bn.bias, bn, 1, 0, False, True,
                            (kh, kw, cin, s, p, reflect, h, w), not bn.training and not torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fan=None):
        xr = _rows(x)
        n, h, w, c = xr.shape
        ho, wo = _out_size(h, 3, 2, 1), _out_size(w, 3, 2, 1)
        y = torch.empty((n, ho, wo, c), dtype=xr.dtype, device=xr.device)
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=xr.device) if ctx.needs_input_grad[0] else None
        with _hip.on_device(xr.device):
            bf = _is_bf16(xr)
            _check(lib().vqseg_maxpool3x3s2_f(bf, 0, _T(xr, "pool input", bf=bf, numel=n * h * w * c), None, n, h, w, c,
                                              _T(y, "pool output", bf=bf, numel=n * ho * wo * c),
                                              _T(idx, "pool argmax", dtype=torch.uint8, numel=n * ho * wo * c), _stream()), "vqseg_maxpool3x3s2_f")
        ctx.save_for_backward(idx)                          # window position of every maximum: the input is not kept
        ctx.cfg = (n, h, w, c, xr.dtype)
        ctx.fan = fan
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, h, w, c, dt = ctx.cfg
        gr = _rows(g).to(dt)
        gx = torch.empty((n, h, w, c), dtype=dt, device=gr.device)
        bf = int(dt == torch.bfloat16)
        add = _fanin_take(ctx.fan, (n, h, w, c), dt)       # the pooled tensor's other consumer deposited its gradient: added in the kernel
        with _hip.on_device(gr.device):
            if add is not None:
                _check(lib().vqseg_maxpool3x3s2_backward_add_f(bf, _T(gr, "pool output gradient", bf=bf, numel=idx.numel()),
                                                               _T(idx, "pool argmax", dtype=torch.uint8), _T(add, "fan-in gradient", bf=bf, numel=n * h * w * c),
                                                               n, h, w, c, _T(gx, "pool input gradient", bf=bf, numel=n * h * w * c), _stream()),
                       "vqseg_maxpool3x3s2_backward_add_f")
            else:
                _check(lib().vqseg_maxpool3x3s2_f(bf, 1, None, _T(gr, "pool output gradient", bf=bf, numel=idx.numel()), n, h, w, c,
                                                  _T(gx, "pool input gradient", bf=bf, numel=n * h * w * c),
                                                  _T(idx, "pool argmax", dtype=torch.uint8), _stream()), "vqseg_maxpool3x3s2_f")
        if ctx.fan is not None and ctx.fan.leftover is not None:
            gx = gx + ctx.fan.leftover.reshape(gx.shape).to(dt)
            ctx.fan.leftover = None
        return _nchw(gx), None


def max_pool_3x3_s2(x):
    if isinstance(x, S3):
        n, h, w, _ = x.rows.shape
        ho, wo = _out_size(h, 3, 2, 1), _out_size(w, 3, 2, 1)
        y = torch.empty((n, ho, wo, 2 * x.c), dtype=torch.bfloat16, device=x.rows.device)
        with _hip.on_device(y.device):
            _check(lib().vqseg_s3_maxpool3x3s2_f(_T(x.rows, "split-3 pool input", bf=2, numel=n * h * w * 2 * x.c), n, h, w, x.c,
                                                 _T(y, "split-3 pool output", bf=2, numel=n * ho * wo * 2 * x.c), _stream()), "vqseg_s3_maxpool3x3s2_f")
        return S3(y, x.c)
    return _MaxPool.apply(x, _fanin_of(x))


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo, align):
        xr = _rows(x)
        n, h, w, c = xr.shape
        y = torch.empty((n, ho, wo, c), dtype=xr.dtype, device=xr.device)
        with _hip.on_device(xr.device):
            bf = _is_bf16(xr)
            _check(lib().vqseg_bilinear_f(bf, 0, _T(xr, "resize input", bf=bf, numel=n * h * w * c), n, h, w, c, ho, wo, int(align),
                                          _T(y, "resize output", bf=bf, numel=n * ho * wo * c), _stream()), "vqseg_bilinear_f")
        ctx.cfg = (n, h, w, c, ho, wo, int(align), xr.dtype)
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        n, h, w, c, ho, wo, align, dt = ctx.cfg
        gr = _rows(g).to(dt)
        gx = torch.empty((n, h, w, c), dtype=dt, device=gr.device)
        with _hip.on_device(gr.device):
            bf = int(dt == torch.bfloat16)
            _check(lib().vqseg_bilinear_f(bf, 1, _T(gr, "resize output gradient", bf=bf, numel=n * ho * wo * c), n, h, w, c, ho, wo, align,
                                          _T(gx, "resize input gradient", bf=bf, numel=n * h * w * c), _stream()), "vqseg_bilinear_f")
        return _nchw(gx), None, None, None


def upsample_bilinear(x, size=None, scale_factor=None, align_corners=False):
    if size is None:
        size = (int(x.shape[-2] * scale_factor), int(x.shape[-1] * scale_factor))
    if isinstance(x, S3):
        n, h, w, _ = x.rows.shape
        y = torch.empty((n, int(size[0]), int(size[1]), 2 * x.c), dtype=torch.bfloat16, device=x.rows.device)
        with _hip.on_device(y.device):
            _check(lib().vqseg_s3_bilinear_f(_T(x.rows, "split-3 resize input", bf=2, numel=n * h * w * 2 * x.c), n, h, w, x.c, int(size[0]), int(size[1]),
                                             int(bool(align_corners)), _T(y, "split-3 resize output", bf=2, numel=y.numel()), _stream()), "vqseg_s3_bilinear_f")
        return S3(y, x.c)
    return _Bilinear.apply(x, int(size[0]), int(size[1]), bool(align_corners))


class _Head1x1(torch.autograd.Function):
    """nn.Conv2d(Cin, num_classes <= 4, 1, bias=False): fp32 logits whatever the activation dtype."""

    @staticmethod
    def forward(ctx, x, weight, fan=None):
        ctx.fan = fan
        xr = _rows(x)
        n, h, w, cin = xr.shape
        cout = weight.shape[0]
        wt = weight.detach().reshape(cout, cin).contiguous()
        y = torch.empty((n, h, w, cout), dtype=torch.float32, device=xr.device)
        with _hip.on_device(xr.device):
            bf = _is_bf16(xr)
            _check(lib().vqseg_head1x1_forward_f(bf, _T(xr, "head input", bf=bf, numel=n * h * w * cin), _f32(wt, "head weight", cout * cin), n * h * w, cin,
                                                 cout, _f32(y, "logits", n * h * w * cout), _stream()), "vqseg_head1x1_forward_f")
        ctx.save_for_backward(xr, wt)
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        xr, wt = ctx.saved_tensors
        n, h, w, cin = xr.shape
        cout = wt.shape[0]
        m = n * h * w
        gr = _rows(g).float()
        gx = torch.empty_like(xr)
        gw = torch.empty((cout, cin), dtype=torch.float32, device=xr.device)
        ws = torch.empty(lib().vqseg_head1x1_backward_workspace_floats(m, cin, cout), dtype=torch.float32, device=xr.device)
        add = _fanin_take(ctx.fan, xr.shape, xr.dtype)      # the prototype loss's gradient of the same decoder output, if it came first
        with _hip.on_device(xr.device):
            bf = _is_bf16(xr)
            _check(lib().vqseg_head1x1_backward_add_f(bf, _T(xr, "head input", bf=bf, numel=m * cin), _f32(wt, "head weight", cout * cin),
                                                      _f32(gr, "logit gradient", m * cout), m, cin, cout, _T(gx, "head input gradient", bf=bf, numel=m * cin),
                                                      _f32(gw, "head weight gradient", cout * cin), _f32(ws, "head workspace"),
                                                      _T(add, "fan-in gradient", bf=bf, numel=m * cin), _stream()), "vqseg_head1x1_backward_add_f")
        if ctx.fan is not None and ctx.fan.leftover is not None:
            gx = gx + ctx.fan.leftover.reshape(gx.shape).to(gx.dtype)
            ctx.fan.leftover = None
        return _nchw(gx), gw.reshape(cout, cin, 1, 1), None


def head_conv1x1(x, weight):
    if isinstance(x, S3) and not torch.is_grad_enabled():    # split-3 eval forward: the head reads hi + lo itself
        n, h, w, _ = x.rows.shape
        cout = weight.shape[0]
        wt = weight.detach().reshape(cout, x.c).contiguous()
        y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.rows.device)
        with _hip.on_device(y.device):
            _check(lib().vqseg_head1x1_forward_f(2, _T(x.rows, "split-3 head input", bf=2, numel=n * h * w * 2 * x.c), _f32(wt, "head weight", cout * x.c),
                                                 n * h * w, x.c, cout, _f32(y, "logits", n * h * w * cout), _stream()), "vqseg_head1x1_forward_f")
        return _nchw(y)
    return _Head1x1.apply(from_s3(x), weight, _fanin_of(x))


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x
        xr = _rows(x)
        y = torch.empty(xr.shape, dtype=dtype, device=xr.device)
        with _hip.on_device(xr.device):
            to_bf = int(dtype == torch.bfloat16)
            _check(lib().vqseg_cast_f(to_bf, _T(xr, "cast input", bf=1 - to_bf), xr.numel(), _T(y, "cast output", bf=to_bf, numel=xr.numel()), _stream()),
                   "vqseg_cast_f")
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        return _Cast.apply(g, ctx.src), None


def cast_act(x, dtype):
    """f32 <-> bf16 activation cast (NHWC), differentiable."""
    if isinstance(x, S3):
        return x if dtype == torch.float32 else cast_act(x.float(), dtype)
    if x.dtype == dtype:
        return x
    if {x.dtype, dtype} != {torch.float32, torch.bfloat16}:
        return x.to(dtype)
    return _Cast.apply(x, dtype)


# ------------------------------------------------------------------------------------------------
# Reliable prototype losses (models/modules/prototype.py): fused forward / backward over the decoder features
# ------------------------------------------------------------------------------------------------
class _ProtoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, proto, labels, keep, conf, variant, scale, margin, easy_margin, fan=None):
        ctx.fan = fan
        xr = _rows(x)
        n, h, w, c = xr.shape
        m, k = n * h * w, proto.shape[0]
        L = lib()
        proto = proto.detach().float().contiguous()
        labels = labels.reshape(-1).long().contiguous()
        keep = keep.reshape(-1).to(torch.uint8).contiguous() if keep is not None else None
        conf = conf.reshape(-1).float().contiguous() if conf is not None else None
        nbytes = L.vqseg_proto_loss_workspace_bytes(m, c, k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xr.device)
        loss = torch.empty((), dtype=torch.float64, device=xr.device)
        with _hip.on_device(xr.device):
            bf = _is_bf16(xr)
            _check(L.vqseg_proto_loss_forward_f(bf, _T(xr, "decoder features", bf=bf, numel=m * c), _f32(proto, "prototypes", k * c),
                                                _T(labels, "labels", dtype=torch.int64, numel=m), _T(keep, "keep mask", dtype=torch.uint8, numel=m),
                                                _f32(conf, "confidence", m), m, c, k, variant, float(scale), float(margin), int(easy_margin),
                                                _T(ws, "workspace", dtype=torch.uint8, numel=nbytes), nbytes,
                                                _T(loss, "loss", dtype=torch.float64, numel=1), _stream()), "vqseg_proto_loss_forward_f")
        ctx.save_for_backward(xr, proto, labels, keep, conf)
        ctx.cfg = (m, c, k, variant, float(scale), float(margin), int(easy_margin))
        return loss

    @staticmethod
    def backward(ctx, g):
        xr, proto, labels, keep, conf = ctx.saved_tensors
        m, c, k, variant, scale, margin, easy = ctx.cfg
        L = lib()
        g32 = g.detach().float().reshape(1).contiguous()
        gx = torch.empty_like(xr)
        gproto = torch.empty_like(proto) if ctx.needs_input_grad[1] else None
        nbytes = L.vqseg_proto_loss_workspace_bytes(m, c, k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xr.device)
        with _hip.on_device(xr.device):
            bf = _is_bf16(xr)
            _check(L.vqseg_proto_loss_backward_f(bf, _T(xr, "decoder features", bf=bf, numel=m * c), _f32(proto, "prototypes", k * c),
                                                 _T(labels, "labels", dtype=torch.int64, numel=m), _T(keep, "keep mask", dtype=torch.uint8, numel=m),
                                                 _f32(conf, "confidence", m), m, c, k, variant, scale, margin, easy, _f32(g32, "loss gradient", 1),
                                                 _T(gx, "feature gradient", bf=bf, numel=m * c), _f32(gproto, "prototype gradient", k * c),
                                                 _T(ws, "workspace", dtype=torch.uint8, numel=nbytes), nbytes, _stream()), "vqseg_proto_loss_backward_f")
        if _fanin_deposit(ctx.fan, gx):                      # the head's backward (same decoder output) adds it in its kernel
            return None, gproto, None, None, None, None, None, None, None, None
        return _nchw(gx), gproto, None, None, None, None, None, None, None, None


def proto_loss_supported(x, num_classes: int) -> bool:
    return x.is_cuda and x.dim() == 4 and x.shape[1] % 8 == 0 and x.shape[1] <= 64 and num_classes <= 4 and \
        x.dtype in (torch.float32, torch.bfloat16)


def proto_loss(x, proto, labels, keep=None, conf=None, variant=1, scale=1.0, margin=0.0, easy_margin=True):
    """-mean( log( exp(S) / (sum_c exp(z_c) + 1e-7) + 1e-7 ) * w ) of the reliable prototype losses (float64 scalar);
    x (N, C, H, W) decoder features, proto (K, C) L2-normalised, labels / keep / conf per pixel in (n, h, w) order."""
    return _ProtoLoss.apply(x, proto, labels, keep, conf, variant, scale, margin, easy_margin, _fanin_of(x))


# ------------------------------------------------------------------------------------------------
# Soft Dice sums (loss/dice_loss.py): inter / sets per (image, class) in one pass, closed-form backward
# ------------------------------------------------------------------------------------------------
class _DiceSums(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, ignore_index, want_ce=False):
        b, c, h, w = pred.shape
        sb, sc, sh, sw = pred.stride()
        if sh != w * sw:                                   # pixels must be linear in memory (NCHW and NHWC both are)
            pred = pred.contiguous()
            sb, sc, sh, sw = pred.stride()
        tgt = target.reshape(b, h * w).long().contiguous()
        L = lib()
        nbytes = L.vqseg_dice_workspace_bytes(b, c, h * w)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
        out = torch.empty(2, b, c, dtype=torch.float32, device=pred.device)
        ce = torch.empty(b, 2, dtype=torch.float32, device=pred.device) if want_ce else None
        ign = -(1 << 62) if ignore_index is None else int(ignore_index)
        with _hip.on_device(pred.device):
            _check(L.vqseg_dice_ce_sums_forward_f(_strided_f32(pred, "logits", b * c * h * w), sb, sc, sw, _T(tgt, "targets", dtype=torch.int64, numel=b * h * w),
                                                  b, c, h * w, ign, _T(ws, "workspace", dtype=torch.uint8, numel=nbytes), nbytes,
                                                  _f32(out[0], "intersections", b * c), _f32(out[1], "set sizes", b * c), _f32(ce, "cross-entropy sums", b * 2),
                                                  _stream()), "vqseg_dice_ce_sums_forward_f")
        ctx.save_for_backward(pred, tgt)
        ctx.cfg = (b, c, h * w, ign, (sb, sc, sw), want_ce)
        if want_ce:
            return out[0], out[1], ce
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_inter, g_sets, g_ce=None):
        pred, tgt = ctx.saved_tensors
        b, c, hw, ign, (sb, sc, sw), want_ce = ctx.cfg
        g = torch.empty_strided(pred.shape, pred.stride(), dtype=torch.float32, device=pred.device)
        gi, gs = g_inter.float().contiguous(), g_sets.float().contiguous()
        gc = g_ce.float().contiguous() if (want_ce and g_ce is not None) else None
        with _hip.on_device(pred.device):
            _check(lib().vqseg_dice_ce_sums_backward_f(_strided_f32(pred, "logits", b * c * hw), sb, sc, sw, _T(tgt, "targets", dtype=torch.int64, numel=b * hw),
                                                       b, c, hw, ign, _f32(gi, "d intersections", b * c), _f32(gs, "d set sizes", b * c),
                                                       _f32(gc, "d cross-entropy sums", b * 2), _strided_f32(g, "logit gradient", b * c * hw), _stream()),
                   "vqseg_dice_ce_sums_backward_f")
        return g, None, None, None


def dice_sums_supported(pred, num_classes: int) -> bool:
    return pred.is_cuda and pred.dim() == 4 and pred.dtype == torch.float32 and 2 <= num_classes <= 4 and pred.shape[1] == num_classes


def dice_sums(pred, target, ignore_index):
    """(inter, sets), each (B, C) float32: sum over pixels of softmax * onehot and of softmax + onehot (dice_loss.py:24-26)."""
    return _DiceSums.apply(pred, target, ignore_index)


def dice_ce_sums(pred, target, ignore_index):
    """(inter, sets, ce): the Dice sums plus ce (B, 2) = (sum of -log softmax[target], number of pixels) over the pixels whose
    target != ignore_index -- nn.CrossEntropyLoss(ignore_index)'s mean is ce[:, 0].sum() / ce[:, 1].sum() -- in the same pass
    over the logits, forward and backward (vqseg_dice_ce_sums_*)."""
    return _DiceSums.apply(pred, target, ignore_index, True)


class _ConvBias(torch.autograd.Function):
    """k x k / stride 1 convolution WITH bias and a handful of output channels, fp32 (the plain Unet's segmentation head: 32 -> 3,
    segmentation_head.py:78-83) on the precise-mode kernels: output channels zero-padded to the kernels' 4-channel granule, the bias in
    the convolution's fused epilogue (vqseg_conv2d_affine_f: scale 1, shift = bias); backward = the data-gradient / weight-gradient
    kernels on the padded gradient, the bias gradient a plain sum."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        xr = _rows(x.float())
        n, h, w, cin = xr.shape
        cout, _, kh, kw = weight.shape
        cp = (cout + 3) // 4 * 4
        dev = xr.device
        wp = torch.zeros(cp, cin, kh, kw, dtype=torch.float32, device=dev)
        wp[:cout] = weight.detach()
        shift = torch.zeros(cp, dtype=torch.float32, device=dev)
        if bias is not None:
            shift[:cout] = bias.detach()
        one = torch.ones(cp, dtype=torch.float32, device=dev)
        L = lib()

        def pack(tr):
            ne = L.vqseg_conv_packed_elems(cp, cin, kh, kw, int(tr))
            hi, lo = torch.empty(ne, dtype=torch.int16, device=dev), torch.empty(ne, dtype=torch.int16, device=dev)
            with _hip.on_device(dev):
                _check(L.vqseg_conv_pack_weights_f32(_f32(wp, "weight", cp * cin * kh * kw), cp, cin, kh, kw, int(tr), _w16(hi, "packed hi", ne),
                                                     _w16(lo, "packed lo", ne), _stream()), "vqseg_conv_pack_weights_f32")
            return hi, lo, ne

        hi, lo, ne = pack(False)
        ho, wo = _out_size(h, kh, 1, pad), _out_size(w, kw, 1, pad)
        y = torch.empty((n, ho, wo, cp), dtype=torch.float32, device=dev)
        with _hip.on_device(dev):
            _check(L.vqseg_conv2d_affine_f(_T(xr, "conv input", bf=0, numel=n * h * w * cin), None, cin, _w16(hi, "packed weights", ne),
                                           _w16(lo, "packed weights (lo)", ne), _f32(one, "unit scale", cp), _f32(shift, "bias", cp), None, 0,
                                           _T(y, "conv output", bf=0, numel=n * ho * wo * cp), n, h, w, cin, cp, kh, kw, 1, pad, 0, ho, wo, 1,
                                           _stream()), "vqseg_conv2d_affine_f")
        ctx.save_for_backward(xr)
        ctx.cfg = (n, h, w, cin, cout, cp, kh, kw, pad, ho, wo, bias is not None)
        ctx.pack = pack
        return y[..., :cout].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        (xr,) = ctx.saved_tensors
        n, h, w, cin, cout, cp, kh, kw, pad, ho, wo, has_bias = ctx.cfg
        dev = xr.device
        L = lib()
        g4 = torch.zeros((n, ho, wo, cp), dtype=torch.float32, device=dev)
        g4[..., :cout] = g.permute(0, 2, 3, 1)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            t_hi, t_lo, _ = ctx.pack(True)                      # [Cin][kh][kw][cp], taps flipped
            gx = _nchw(_conv_raw(g4, None, cp, t_hi, t_lo, (n, h, w, cin), None, n, ho, wo, cp, cin, kh, kw, 1, kh - 1 - pad, False, 1, h, w))
        if ctx.needs_input_grad[1]:
            nbytes = L.vqseg_conv2d_wgrad_workspace_bytes(n, h, w, cin, ho, wo, cp, kh, kw)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            gw4 = torch.empty((cp, cin, kh, kw), dtype=torch.float32, device=dev)
            with _hip.on_device(dev):
                _check(L.vqseg_conv2d_wgrad_f(_T(g4, "conv output gradient", bf=0, numel=n * ho * wo * cp), _T(xr, "conv input", bf=0, numel=n * h * w * cin),
                                              None, cin, n, h, w, cin, ho, wo, cp, kh, kw, 1, pad, 0, 1, cin, 0, 0,
                                              _T(ws, "wgrad workspace", dtype=torch.uint8, numel=nbytes), nbytes, _f32(gw4, "weight gradient", cp * cin * kh * kw),
                                              _stream()), "vqseg_conv2d_wgrad_f")
            gw = gw4[:cout]
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 2, 3))
        return gx, gw, gb, None


def conv2d_bias(x, weight, bias, padding: int):
    """F.conv2d(x.float(), weight, bias, stride 1, padding) on the HIP kernels (few output channels; fp32 like the reference's head)."""
    return _ConvBias.apply(x, weight, bias, int(padding))


class _CPSCombine(torch.autograd.Function):
    """vqseg_cps_loss_combine_f: total = sup terms + cps_w * cps terms + commitment + prototype from the Dice (+ CE) sums, in one launch,
    with the gradient of the total with respect to every input written in the same launch (backward = one scaling)."""

    @staticmethod
    def forward(ctx, spec, *ts):
        n_sup, n_cps, has_ce, cps_w, ce_w, eps, commit_w, proto_w, n_commit, n_proto = spec
        nt, per = n_sup + n_cps, (3 if has_ce else 2)
        terms = [ts[i * per:(i + 1) * per] for i in range(nt)]
        commits = ts[nt * per: nt * per + n_commit]
        protos = ts[nt * per + n_commit:]
        dev = terms[0][0].device
        c = terms[0][0].shape[1]
        bs = [int(t[0].shape[0]) for t in terms]
        levels = int(commits[0].numel()) if commits else 0
        sizes = []
        for b in bs:
            sizes += [b * c, b * c] + ([b * 2] if has_ce else [])
        gbuf = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        views = list(torch.split(gbuf, sizes))
        out = torch.empty(4 + nt, dtype=torch.float32, device=dev)
        P4 = ctypes.c_void_p * 4

        def arr(tensors, what, dtype=torch.float32):
            a = P4()
            for i, t in enumerate(tensors):
                a[i] = _T(t, what, dtype=dtype) if t is not None else None
            return a

        inter = arr([t[0] for t in terms], "intersections")
        sets = arr([t[1] for t in terms], "set sizes")
        ce = arr([t[2] for t in terms], "cross-entropy sums") if has_ce else None
        g_inter = arr(views[0::per], "gradient of the intersections")
        g_sets = arr(views[1::per], "gradient of the set sizes")
        g_ce = arr(views[2::per], "gradient of the cross-entropy sums") if has_ce else None
        b_arr = (ctypes.c_int * 4)(*(bs + [0] * (4 - nt)))
        with _hip.on_device(dev):
            _check(lib().vqseg_cps_loss_combine_f(n_sup, n_cps, c, inter, sets, ce, b_arr, float(cps_w), float(ce_w), float(eps),
                                                  arr(commits, "commitment losses"), n_commit, levels, float(commit_w),
                                                  arr(protos, "prototype losses", dtype=torch.float64), n_proto, float(proto_w),
                                                  g_inter, g_sets, g_ce, _f32(out, "loss terms", 4 + nt), _stream()), "vqseg_cps_loss_combine_f")
        ctx.gbuf, ctx.sizes, ctx.shapes = gbuf, sizes, [tuple(x.shape) for t in terms for x in t]
        ctx.tail = (n_commit, n_proto, levels, float(commit_w), float(proto_w), [tuple(x.shape) for x in commits])
        total = out[0].clone()
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, g, _g_stats):
        n_commit, n_proto, levels, commit_w, proto_w, cshapes = ctx.tail
        gb = ctx.gbuf * g
        grads = [v.reshape(sh) for v, sh in zip(torch.split(gb, ctx.sizes), ctx.shapes)]
        gc = (g * commit_w)
        gp = (g.double() * proto_w)
        return (None, *grads, *[gc.expand(sh) for sh in cshapes], *[gp for _ in range(n_proto)])


def cps_loss_combine(sup_terms, cps_terms, cps_weight, ce_weight, commits, commit_weight, protos, proto_weight, eps=1e-6):
    """-> (total, stats) with stats = [total, commitment, prototype, cps_1 + cps_2, sup_1, .., cps_1, ..] (no gradient through stats).
    `sup_terms` / `cps_terms`: lists of (inter, sets) or (inter, sets, ce) from dice_sums / dice_ce_sums; `commits`: fp32 vectors of
    one length; `protos`: float64 scalars.  None when the inputs are not what the kernel takes (the caller combines with torch ops)."""
    terms = list(sup_terms) + list(cps_terms)
    if not terms or len(terms) > 4 or len(commits) > 4 or len(protos) > 4 or not py_opt("py_loss_combine", 1):
        return None
    has_ce = len(terms[0]) == 3
    flat = []
    for t in terms:
        if len(t) != (3 if has_ce else 2) or not all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() for x in t):
            return None
        if t[0].dim() != 2 or t[0].shape != t[1].shape or t[0].shape[1] != terms[0][0].shape[1] or (has_ce and tuple(t[2].shape) != (t[0].shape[0], 2)):
            return None
        flat += list(t)
    if not all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 1 and x.shape == commits[0].shape for x in commits):
        return None
    if not all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float64 and x.numel() == 1 for x in protos):
        return None
    spec = (len(sup_terms), len(cps_terms), has_ce, float(cps_weight), float(ce_weight), float(eps), float(commit_weight), float(proto_weight),
            len(commits), len(protos))
    return _CPSCombine.apply(spec, *flat, *commits, *protos)


def softmax_stats(logits, want_label=True, want_entropy=True, want_top=False):
    """(label (B, H, W) i64, entropy (B, H, W) f32, top-probability (B, H, W) f32) of softmax(logits, dim=1) in one HIP pass
    (None for outputs not wanted).  No gradient flows (the callers detach / use them as masks and targets)."""
    x = logits.detach()
    if x.dtype != torch.float32:
        x = x.float()
    b, c, h, w = x.shape
    sb, sc, sh, sw = x.stride()
    if sh != w * sw:
        x = x.contiguous()
        sb, sc, sh, sw = x.stride()
    dev = x.device
    label = torch.empty((b, h, w), dtype=torch.int64, device=dev) if want_label else None
    ent = torch.empty((b, h, w), dtype=torch.float32, device=dev) if want_entropy else None
    top = torch.empty((b, h, w), dtype=torch.float32, device=dev) if want_top else None
    with _hip.on_device(dev):
        _check(lib().vqseg_softmax_stats_f(_strided_f32(x, "logits", b * c * h * w), sb, sc, sw, b, c, h * w, _T(label, "labels", dtype=torch.int64, numel=b * h * w),
                                           _f32(ent, "entropy", b * h * w), _f32(top, "top probability", b * h * w), _stream()), "vqseg_softmax_stats_f")
    return label, ent, top


def percentile(values, percent: float):
    """np.percentile(values, percent) (linear interpolation) as a 0-d device tensor, for any number of fp32 values below 2^32:
    the two order statistics around the virtual index come from an exact radix select (vqseg_order_stats_f) instead of a
    full sort, and the index is taken in double as numpy does (torch.quantile rounds it to fp32)."""
    x = values.detach().reshape(-1)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    n = x.numel()
    virtual = percent / 100.0 * (n - 1)
    k = min(max(int(virtual), 0), n - 1)
    ws = torch.empty(lib().vqseg_order_stats_workspace_bytes(), dtype=torch.uint8, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    with _hip.on_device(x.device):
        _check(lib().vqseg_order_stats_f(_f32(x, "values", n), n, k, _T(ws, "workspace", dtype=torch.uint8), ws.numel(), _f32(out, "order statistics", 2),
                                         _stream()), "vqseg_order_stats_f")
    frac = virtual - k
    if frac <= 0.0:
        return out[0]
    return torch.lerp(out[0], out[1], frac)


def softmax_stats_supported(logits) -> bool:
    return logits.is_cuda and logits.dim() == 4 and 2 <= logits.shape[1] <= 4

