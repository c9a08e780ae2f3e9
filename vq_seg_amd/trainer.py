"""Cross-pseudo-supervision (CPS) training of two VQ-UNets: the caller of the hot path.

Mirrors the loop bodies of the reference's trainers
  v2: train_vqreptunet1x1v2.py:137-211  (score-mask CPS, CE + Dice, confidence threshold)
  v1: deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203 (entropy-percentile pseudo labels,
      criterion from cfg, `percent` schedule)
as one `CPSTrainer.step()`; data-parallel across the GPUs of a node with a bucketed gradient
all-reduce over RCCL that overlaps the backward pass (no reference counterpart: SURVEY 2, 8e).
wandb / image dumps of the reference trainer are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import contextlib
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import dist as vdist
from .loss import make_loss
from .loss.dice_loss import DiceLoss, ce_dice_loss
from .measurement import confusion_matrix_device, miou_device
from . import nnf
from .models import init_weight
from .models.networks import make_model
from .optim import HipAdam
from .utils.ckpoints import load_training_state, restore_initted, save_ckpoints
from .utils.lr_schedulers import CosineAnnealingLR


# ----------------------------------------------------------------------------------------------
# gradient all-reduce: flat fp32 buckets, launched from post-accumulate hooks while backward runs
# ----------------------------------------------------------------------------------------------
class GradBuckets:
    """Owns the .grad storage of `params` as views into a few flat buffers ("buckets", filled in
    reverse parameter order = the order backward produces them).  With world_size > 1 each bucket is
    all-reduced (sum, then scaled by 1/world) as soon as all its gradients have been accumulated."""

    def __init__(self, params: List[nn.Parameter], bucket_mb: float = 64.0):
        self.params = [p for p in params if p.requires_grad]
        self.world = vdist.world_size()
        self.on = vdist.collectives_on()                   # world > 1, or the single-rank RCCL drive (VQSEG_DIST_SINGLE)
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets: List[torch.Tensor] = []
        self.owner: Dict[int, int] = {}
        groups, cur, cur_n = [], [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for bi, g in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in g), dtype=torch.float32, device=g[0].device)
            off = 0
            for p in g:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.owner[id(p)] = bi
            self.buckets.append(flat)
        self._sizes = [len(g) for g in groups]
        self._pending = list(self._sizes)
        self._launched = [False] * len(groups)
        self._handles = []
        self._reported = set()
        # parameters that never report a gradient (the frozen codebooks -- vq_img.py:236-239 -- and, in the v1 model, the prototypes
        # that enter through `.data`, prototype.py:556) would keep their buckets from ever completing inside backward: they are
        # learnt on the first step (whoever has not reported by finish()) and left out of the count from then on
        self._silent = None
        self._names = {}
        self.launched_in_backward = []                     # per step: which buckets were reduced from inside backward (tests)
        self.launch_sequence: List[int] = []               # the order this step's bucket all-reduces were issued in (must be the same on
                                                           # every rank: bench.py / tests compare it across ranks)
        # RCCL has an averaging reduction; gloo (CPU tests, rehearsal) does not: sum, then scale
        self._avg = dist.ReduceOp.AVG if (self.on and dist.get_backend() == "nccl") else None
        self.producer_streams = []          # set by CPSTrainer: the side stream(s) the gradient kernels of these params run on
        for p in self.params:
            # the HIP weight-gradient kernels add straight into the bucket views (nnf grad sinks) and report here;
            # parameters whose gradient still comes from autograd (VQ-free torch ops) report through the hook
            p._vq_grad_sink = self._on_grad if self.on else None
            if self.on:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def zero(self):
        for b in self.buckets:
            b.zero_()
        for p in self.params:
            p._vq_uses = 0
        self._pending = list(self._sizes)
        if self._silent:
            for pid in self._silent:
                self._pending[self.owner[pid]] -= 1
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._reported = set()
        self.launch_sequence = []

    def _launch(self, bi):
        self._launched[bi] = True
        self.launch_sequence.append(bi)
        # The collective orders itself after the CURRENT stream only.  The last gradient of a bucket may report from an
        # autograd hook running on another stream than the one the HIP weight-gradient kernels wrote the rest of the bucket
        # on, so order the current stream after everything those streams hold so far (all of this bucket's gradients).
        if self.producer_streams and self.buckets[bi].is_cuda:
            cur = torch.cuda.current_stream(self.buckets[bi].device)
            for s in self.producer_streams:
                if s != cur:
                    cur.wait_stream(s)
        self._handles.append(dist.all_reduce(self.buckets[bi], op=self._avg or dist.ReduceOp.SUM, async_op=True))

    def _on_grad(self, p):
        # A sunk parameter reports through its sink (last contribution) AND, later, through the post-accumulate hook
        # (autograd runs AccumulateGrad for it with an undefined gradient): count every parameter once per step.
        if id(p) in self._reported:
            return
        self._reported.add(id(p))
        bi = self.owner[id(p)]
        if self._silent and id(p) in self._silent:
            if self._launched[bi]:
                raise RuntimeError("a parameter classified as gradient-free on the first step produced a gradient after its bucket had "
                                   "been reduced; call GradBuckets.relearn() when the set of trained parameters changes")
            self._silent.discard(id(p))                    # it does report after all: count it again from the next step on
            return
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and not self._launched[bi]:
            self._launch(bi)

    def finish(self):
        """Call after backward: reduce what is left (parameters without a gradient this step keep
        their bucket from completing), wait, average."""
        if not self.on:
            return
        self.launched_in_backward = list(self._launched)
        if self._silent is None:
            self._silent = {id(p) for p in self.params if id(p) not in self._reported}
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        for h in self._handles:
            h.wait()
        if self._avg is None:                              # gloo: no averaging reduction
            for b in self.buckets:
                b.mul_(1.0 / self.world)

    def relearn(self):
        """forget which parameters are gradient-free (after freezing / unfreezing parts of the model)"""
        self._silent = None


# ----------------------------------------------------------------------------------------------
# synthetic data (no dataset is available offline; shapes and value ranges of data/dataset.py:41-62)
# ----------------------------------------------------------------------------------------------
class SyntheticCropWeed:
    """Images: learnable colour-coded blobs + noise in [0,1] (TF.to_tensor range, no mean/std
    normalisation, q14); labels: class ids 0/1/2 (post `img_to_label`).  Deterministic per (seed, rank)."""

    def __init__(self, size: int, batch: int, device, seed: int = 42, cell: int = 32, num_classes: int = 3):
        self.size, self.batch, self.device, self.cell, self.nc = size, batch, device, cell, num_classes
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed * 1000 + vdist.rank())
        self.palette = torch.tensor([[0.25, 0.20, 0.15], [0.20, 0.55, 0.25], [0.55, 0.60, 0.20]])

    def _one(self):
        low = max(self.size // self.cell, 1)
        lab = torch.randint(0, self.nc, (self.batch, 1, low, low), generator=self.gen).float()
        lab = F.interpolate(lab, size=(self.size, self.size), mode="nearest")[:, 0].long()
        img = self.palette[lab].permute(0, 3, 1, 2) + 0.15 * torch.rand(self.batch, 3, self.size, self.size, generator=self.gen)
        return img.clamp_(0, 1), lab

    def labelled(self):
        img, lab = self._one()
        return img.to(self.device, non_blocking=True), lab.to(self.device, non_blocking=True)

    def unlabelled(self):
        return self._one()[0].to(self.device, non_blocking=True)


# ----------------------------------------------------------------------------------------------
@dataclass
class CPSConfig:
    model: dict
    recipe: str = "v1"                       # "v1": entropy-percentile CPS; "v2": score-mask CPS
    num_classes: int = 3
    learning_rate: float = 1e-4
    min_lr: float = 1e-7
    total_iters: int = 1000
    warmup_steps: int = 0
    cps_loss_weight: float = 1.0
    total_commitment_loss_weight: float = 1.0
    total_prototype_loss_weight: float = 0.01
    unsup_loss_drop_percent: float = 20.0
    confidence_threshold: float = 0.7
    criterion: str = "dice_loss"
    init_weights: bool = True
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    amp_dtype: Optional[torch.dtype] = None   # torch.bfloat16 = the ROCm-native counterpart of the reference's fp16 AMP
    eval_amp: bool = False                    # False = the reference: the two no-grad pseudo-label forwards run OUTSIDE autocast, in
                                              # fp32 (train_vqreptunet1x1v2.py:143-149; deprecated/train_with_test_pt_pseudo_entropy_
                                              # reg.py:150-156); True = an all-bf16 step (a build-side speed mode, NOT the reference's)
    keep_aux: bool = False                    # keep the step's pseudo-label masks / scores in `CPSTrainer.aux` (parity tests)
    bucket_mb: float = 64.0
    two_streams: bool = True                  # each network of the pair on its own HIP stream (see CPSTrainer.__init__)
    wgrad_side_stream: bool = False           # opt-in: weight-gradient kernels on a side stream per network (nnf.WGRAD_SIDE_STREAMS); measured +-0 (r3)
    seed: int = 42
    extra: dict = field(default_factory=dict)


def broadcast_module_state(m: nn.Module, src: int = 0) -> None:
    """rank `src`'s parameters and buffers to every rank: ONE broadcast per dtype over a flat copy (r3 issued ~750 per-tensor
    broadcasts per network)."""
    by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
    for t in list(m.parameters()) + list(m.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def regularized_pseudo_label(raw: torch.Tensor, percent: float) -> torch.Tensor:
    """make_regularized_pseudo_label (deprecated/train_with_test_pt_pseudo_entropy_reg.py:30-39) with
    the percentile taken on the device (np.percentile's linear interpolation between two order statistics)."""
    if nnf.softmax_stats_supported(raw):
        label, entropy, _ = nnf.softmax_stats(raw)          # one HIP pass (vqseg_softmax_stats_f)
    else:
        prob = torch.softmax(raw.float(), dim=1)
        label = torch.argmax(prob, dim=1)
        entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
    if entropy.is_cuda:
        thresh = nnf.percentile(entropy, percent)         # exact radix select (vqseg_order_stats_f), no sort, no size limit
    else:
        thresh = torch.quantile(entropy.detach().flatten(), percent / 100.0)
    return label.masked_fill_(entropy >= thresh, 255)         # (`label` is this function's own tensor: in place, one pass less than where + full_like)


def _argmax_classes(score: torch.Tensor) -> torch.Tensor:
    """torch.argmax(score, dim=1) (first maximum) -- on the device through the pseudo-label statistics kernel (one pass at memory speed;
    ATen's strided reduction over the class dimension takes ~200 us for 32 x 3 x 512 x 512 logits, this ~35)"""
    if nnf.softmax_stats_supported(score):
        return nnf.softmax_stats(score, want_label=True, want_entropy=False, want_top=False)[0]
    return torch.argmax(score, dim=1)


def score_mask(pred: torch.Tensor, pseudo: torch.Tensor, th: float = 0.7) -> torch.Tensor:
    """train_vqreptunet1x1v2.py:43-46."""
    if nnf.softmax_stats_supported(pred):
        top = nnf.softmax_stats(pred, want_label=False, want_entropy=False, want_top=True)[2]
    else:
        top = torch.softmax(pred.float(), dim=1).max(dim=1)[0]
    return torch.where(top > th, pseudo, torch.full_like(pseudo, 255))


class CPSTrainer:
    def __init__(self, cfg: CPSConfig, device, models: Optional[List[nn.Module]] = None):
        """`models`: an already built (model_1, model_2) pair on `device` (fixtures with given weights); else the pair is built
        from cfg.model as the reference trainer does (train_vqreptunet1x1v2.py:70-80)."""
        self.cfg, self.device = cfg, device
        self.aux: Dict[str, torch.Tensor] = {}
        torch.manual_seed(cfg.seed)                      # same initial weights on every rank
        self.models = list(models) if models is not None else [make_model(cfg.model).to(device), make_model(cfg.model).to(device)]
        if cfg.init_weights and models is None:          # train_vqreptunet1x1v2.py:73-80
            for m in self.models:
                init_weight([m.decoder, m.segmentation_head], nn.init.kaiming_normal_, nn.BatchNorm2d, cfg.bn_eps,
                            cfg.bn_momentum, mode="fan_in", nonlinearity="relu")
        if vdist.collectives_on():
            for m in self.models:
                broadcast_module_state(m)
                nnf.invalidate_weight_caches(m)          # `.data` writes do not bump the version counter (_wcache)
        for m in self.models:
            m.async_code_usage = True                     # no host sync inside forward (usage is read after the step)
        self.buckets = [GradBuckets(list(m.parameters()), cfg.bucket_mb) for m in self.models]
        # the two networks of a CPS pair are independent until the loss: each runs on its own HIP stream so that one
        # model's small, latency-bound kernels (statistics merges, reductions, packs) hide under the other's large ones
        import os as _os
        self._two_streams = device.type == "cuda" and cfg.two_streams and _os.environ.get("VQSEG_TWO_STREAMS", "1") == "1"
        prio = nnf.py_opt("py_stream_prio", 0) if device.type == "cuda" else 0      # A/B: 1 = the networks' streams above the weight-gradient side streams
        self._streams = [torch.cuda.Stream(device, priority=-1 if prio == 1 else 0) for _ in self.models] if self._two_streams else []
        # weight-gradient kernels (+ their slab sums) run on a side stream per network (nnf.WGRAD_SIDE_STREAMS): nothing later in
        # backward depends on them, so they fill the chip under the latency-bound links of the main chain.  They add into the
        # buckets, so the side streams are producer streams of the buckets and are joined before the optimiser step.
        self._wgrad_streams = []
        if self._two_streams and cfg.wgrad_side_stream and nnf.py_opt("py_wgrad_side", 1):
            self._wgrad_streams = [torch.cuda.Stream(device, priority=-1 if prio == 2 else 0) for _ in self.models]
        for b, s_ in zip(self.buckets, self._streams):
            b.producer_streams = [s_]
        for b, s_, w_ in zip(self.buckets, self._streams, self._wgrad_streams):
            b.producer_streams.append(w_)
        self._pending_sides = set()
        # the reference's optimiser (train_vqreptunet1x1v2.py:106-107) on the HIP step kernel: Adam + the convolution kernels' weight
        # images in one launch per network (optim.HipAdam IS a torch.optim.Adam: same state / state_dict layout)
        self.opts = [HipAdam(m.parameters(), lr=cfg.learning_rate, betas=(0.9, 0.999)) if self.device.type == "cuda" and nnf.py_opt("py_hip_adam", 1)
                     else torch.optim.Adam(m.parameters(), lr=cfg.learning_rate, betas=(0.9, 0.999), fused=self.device.type == "cuda")
                     for m in self.models]
        self.sched = CosineAnnealingLR(cfg.learning_rate, cfg.min_lr, cfg.total_iters, cfg.warmup_steps)
        self.ce = nn.CrossEntropyLoss(ignore_index=255)
        self.criterion = make_loss(cfg.criterion, cfg.num_classes, ignore_index=255)
        self.iter = 0

    # -- one model forward under the configured precision
    def _fwd(self, model, *a, **kw):
        side = self._side_stream(model)
        if side is None:
            return self._fwd_here(model, *a, **kw)
        main = torch.cuda.current_stream()
        side.wait_stream(main)                                          # inputs / zeroed buckets come from the caller's stream
        with torch.cuda.stream(side):
            out = self._fwd_here(model, *a, **kw)
        for t in out:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)                                   # consumed on the caller's stream
        self._pending_sides.add(side)
        return out

    def _fwd_pair(self, a1, a2, use_amp=True, **kw):
        """model 1 on a1 = (x[, gt]) and model 2 on a2, phase by phase: encoders overlap on the two streams, the VQ phases
        run one after the other with the other stream idle (the distance kernels fill the GPU on their own, and their
        in-stream timing -- bench.py's roofline -- then measures the kernel, not the sharing), decoders overlap again.
        `use_amp=False`: this pair runs outside autocast whatever cfg.amp_dtype says (the pseudo-label passes)."""
        m1, m2 = self.models
        amp = self.cfg.amp_dtype if use_amp else None

        def on(stream, fn, *args, **kws):
            if stream is None:
                if amp is None:
                    return fn(*args, **kws)
                with torch.autocast("cuda", dtype=amp):
                    return fn(*args, **kws)
            with torch.cuda.stream(stream):
                if amp is None:
                    return fn(*args, **kws)
                with torch.autocast("cuda", dtype=amp):
                    return fn(*args, **kws)

        if not self._two_streams:                                       # same phase order (also the order of the RNG draws
            f1, f2 = on(None, m1.encode, a1[0]), on(None, m2.encode, a2[0])   # of the k-means inits) on one stream
            q1, q2 = on(None, m1.quantize, f1), on(None, m2.quantize, f2)
            return on(None, m1.finish, *q1, *a1[1:], **kw), on(None, m2.finish, *q2, *a2[1:], **kw)
        main = torch.cuda.current_stream()
        s1, s2 = self._streams
        s1.wait_stream(main)
        s2.wait_stream(main)
        f1 = on(s1, m1.encode, a1[0])
        f2 = on(s2, m2.encode, a2[0])
        alone = nnf.py_opt("py_vq_alone", 1)                            # 0 (A/B runs): the VQ phases share the chip like everything else
        if alone:
            s1.wait_event(s2.record_event())                            # model 2's encoder done before model 1 quantises
        q1 = on(s1, m1.quantize, f1)
        if alone:
            s2.wait_event(s1.record_event())
        q2 = on(s2, m2.quantize, f2)
        if alone:
            s1.wait_event(s2.record_event())                            # ... and model 1 stays idle meanwhile
        o1 = on(s1, m1.finish, *q1, *a1[1:], **kw)
        o2 = on(s2, m2.finish, *q2, *a2[1:], **kw)
        for t in tuple(o1) + tuple(o2):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        self._pending_sides.update((s1, s2))
        return o1, o2

    def _join(self):
        """the caller's stream waits for everything queued on the per-model streams"""
        main = torch.cuda.current_stream()
        for s_ in self._pending_sides:
            main.wait_stream(s_)
        self._pending_sides.clear()

    def _wgrad_sides(self, on: bool):
        """register / unregister the weight-gradient side streams with nnf for the duration of a backward pass"""
        for s_, w_ in zip(self._streams, self._wgrad_streams):
            if on and self._two_streams:
                nnf.WGRAD_SIDE_STREAMS[s_.cuda_stream] = w_
            else:
                nnf.WGRAD_SIDE_STREAMS.pop(s_.cuda_stream, None)

    def _side_stream(self, model):
        if not self._two_streams:
            return None
        return self._streams[self.models.index(model)]

    def _fwd_here(self, model, *a, **kw):
        if self.cfg.amp_dtype is None:
            return model(*a, **kw)
        with torch.autocast("cuda", dtype=self.cfg.amp_dtype):
            return model(*a, **kw)

    def step(self, l_input, l_target, ul_input, epoch_frac: float = 0.0) -> Dict[str, torch.Tensor]:
        """One CPS iteration on B labelled + B unlabelled images (6 forwards, 4 backwards per pair of
        models).  Returns device scalars (no host sync)."""
        cfg = self.cfg
        m1, m2 = self.models
        if self._two_streams and l_input.is_cuda and nnf.py_opt("py_loss_streams", 1) == 1:
            # each network's buckets are written by ITS stream's backward: zeroed there too, under the first forward's kernels
            main = torch.cuda.current_stream()
            for b, s_ in zip(self.buckets, self._streams):
                s_.wait_stream(main)
                with torch.cuda.stream(s_):
                    b.zero()
                self._pending_sides.add(s_)
        else:
            for b in self.buckets:
                b.zero()
        nnf.drop_pending_wgrads()
        if l_input.is_cuda:
            # one layout conversion per batch instead of one per forward, and one stem patch matrix per batch for the six
            # forwards of the step (nnf.stem_share_*): the two networks and the two passes over the unlabelled batch see
            # the same images
            l_input = l_input.contiguous(memory_format=torch.channels_last)
            ul_input = ul_input.contiguous(memory_format=torch.channels_last)
            if nnf.py_opt("py_stem_share", 1):
                nnf.stem_share_begin()
        prev = nnf.set_fanin_fusion(True)                              # fan-in adds inside the consumers' kernels; checked after backward
        try:
            return self._step(l_input, l_target, ul_input, epoch_frac)
        finally:
            nnf.set_fanin_fusion(prev)
            nnf.stem_share_end()

    def _step(self, l_input, l_target, ul_input, epoch_frac):
        cfg = self.cfg
        m1, m2 = self.models
        # r4: the glue between the forwards -- pseudo-label targets, the loss block -- runs on the two networks' OWN streams (what
        # network k's stream produced is consumed there; the one exchange is the other network's mask / target, an event each way)
        # instead of joining both streams into the caller's and running ~150 small kernels there with the chip idle (measured on a
        # kernel trace: 3.2 ms of a 152 ms step).  Same operations, same operands, same order per network.
        split = self._two_streams and nnf.py_opt("py_loss_streams", 1) == 1
        streams = self._streams if split else (None, None)
        main = torch.cuda.current_stream() if l_input.is_cuda else None

        def on(k):
            return torch.cuda.stream(streams[k]) if split else contextlib.nullcontext()

        def exchange(t_from_1, t_from_2):
            """network 1's stream may read t_from_2 (made on network 2's stream) and vice versa"""
            if split:
                e1, e2 = streams[0].record_event(), streams[1].record_event()
                streams[0].wait_event(e2)
                streams[1].wait_event(e1)
                t_from_2.record_stream(streams[0])
                t_from_1.record_stream(streams[1])

        with torch.no_grad():                                           # pseudo labels from eval passes
            m1.eval(); m2.eval()
            o1, o2 = self._fwd_pair((ul_input,), (ul_input,), use_amp=cfg.eval_amp)
            score_1, score_2 = o1[0], o2[0]
            if not split:
                self._join()
            with on(0):
                score_1 = score_1.float()
            with on(1):
                score_2 = score_2.float()
            m1.train(); m2.train()
        if cfg.recipe == "v1":
            percent = 100 - cfg.unsup_loss_drop_percent * (1 - epoch_frac)
            kw = dict(percent=percent)
            with on(1):
                gt_ul_1 = _argmax_classes(score_2)
            with on(0):
                gt_ul_2 = _argmax_classes(score_1)
        else:
            kw = dict(th=cfg.confidence_threshold)
            gt_ul_1, gt_ul_2 = score_2, score_1
        if split:                                                       # targets ready: the events the OTHER network's unlabelled forward waits for
            e_gt = (streams[1].record_event(), streams[0].record_event())       # (gt_ul_1 on network 2's stream, gt_ul_2 on network 1's)
        (ps1, c_l1, _u, p_l1), (ps2, c_l2, _u, p_l2) = self._fwd_pair((l_input, l_target), (l_input, l_target), **kw)
        if split:
            streams[0].wait_event(e_gt[0])
            streams[1].wait_event(e_gt[1])
            gt_ul_1.record_stream(streams[0])
            gt_ul_2.record_stream(streams[1])
        (pu1, c_u1, _u, p_u1), (pu2, c_u2, usage, p_u2) = self._fwd_pair((ul_input, gt_ul_1), (ul_input, gt_ul_2), **kw)
        if not split:
            self._join()
        crit = self.criterion if cfg.recipe == "v1" else self._ce_dice

        def own_mask(ps, pu):
            ps, pu = ps.float(), pu.float()
            pred = torch.cat([ps, pu], dim=0)
            if cfg.recipe == "v1":
                return ps, pu, pred, regularized_pseudo_label(pred, percent)
            return ps, pu, pred, score_mask(pred, _argmax_classes(pred), cfg.confidence_threshold)

        with on(0):
            ps1, pu1, pred_1, mask_1 = own_mask(ps1, pu1)
        with on(1):
            ps2, pu2, pred_2, mask_2 = own_mask(ps2, pu2)
        exchange(mask_1, mask_2)
        # r4: with the Dice criterion (no class weights) the four terms stay as their (inter, sets[, ce]) sums and the whole combination
        # -- terms, commitment, prototype, total, and its gradient -- is one launch (nnf.cps_loss_combine); else scalar torch ops
        crt = self.criterion
        fuse = (isinstance(crt, DiceLoss) and crt.weight is None and crt.ignore_index is not None and pred_1.is_cuda
                and nnf.dice_sums_supported(pred_1, crt.num_classes) and nnf.py_opt("py_loss_combine", 1) == 1)
        if fuse:
            def crit(pred, target):                                     # noqa: F811  (the sums; combined below)
                if cfg.recipe == "v1":
                    return nnf.dice_sums(pred, target, crt.ignore_index)
                return nnf.dice_ce_sums(pred, target, crt.ignore_index)
        with on(0):
            cps_1, sup_1 = crit(pred_1, mask_2), crit(ps1, l_target)
        with on(1):
            cps_2, sup_2 = crit(pred_2, mask_1), crit(ps2, l_target)
        if split:
            self._pending_sides.update(streams)
            self._join()
            for t in (cps_1, sup_1, cps_2, sup_2, ps1, pu2, mask_1, mask_2, score_1, score_2):
                for x in (t if isinstance(t, tuple) else (t,)):
                    x.record_stream(main)
        if cfg.keep_aux:
            self.aux = dict(mask_1=mask_1, mask_2=mask_2, score_1=score_1, score_2=score_2, pred_sup_1=ps1.detach(), pred_ul_2=pu2.detach())
        lr = self.sched.get_lr(self.iter)
        for o in self.opts:
            o.param_groups[0]["lr"] = lr
        combined = None
        if fuse:
            combined = nnf.cps_loss_combine([sup_1, sup_2], [cps_1, cps_2], cfg.cps_loss_weight, 0.0 if cfg.recipe == "v1" else 0.5,
                                            [c_l1, c_l2, c_u1, c_u2], cfg.total_commitment_loss_weight,
                                            [p_l1, p_l2, p_u1, p_u2], cfg.total_prototype_loss_weight)
            if combined is None:                                        # inputs the kernel does not take: the same terms with torch ops
                def scalar(t):
                    dice = 1 - (2 * t[0] / (t[1] + 1e-6)).mean(dim=0).mean()
                    return dice if len(t) == 2 else 0.5 * (t[2][:, 0].sum() / t[2][:, 1].sum()) + dice
                cps_1, sup_1, cps_2, sup_2 = scalar(cps_1), scalar(sup_1), scalar(cps_2), scalar(sup_2)
        if combined is not None:
            loss, stats = combined
            com_sum, prototype, cps, sup_1, sup_2 = stats[1], stats[2], stats[3], stats[4], stats[5]
        else:
            cps = cps_1 + cps_2
            commitment = (c_l1 + c_l2 + c_u1 + c_u2) * cfg.total_commitment_loss_weight
            prototype = (p_l1 + p_l2 + p_u1 + p_u2) * cfg.total_prototype_loss_weight
            com_sum = commitment.sum()
            loss = sup_1 + sup_2 + cfg.cps_loss_weight * cps + com_sum + prototype.float()
        self._wgrad_sides(True)
        try:
            loss.backward()
        finally:
            self._wgrad_sides(False)
        nnf.flush_pending_wgrads()                                      # two-use weight gradients whose second use never came (none, normally)
        nnf.check_fanin_consumed()
        if self._two_streams and nnf.py_opt("py_opt_streams", 1):
            # each network's gradient reduction + optimiser step on ITS stream (the sinks wrote p.grad there): the two Adam launches
            # overlap each other and the metric kernels below; the caller's stream joins both before the step returns
            main = torch.cuda.current_stream()
            for i, (b, o, s_) in enumerate(zip(self.buckets, self.opts, self._streams)):
                s_.wait_stream(main)                                    # backward's leaf streams are joined into `main` by autograd
                if self._wgrad_streams:
                    s_.wait_stream(self._wgrad_streams[i])
                with torch.cuda.stream(s_):
                    b.finish()
                    o.step()
                self._pending_sides.add(s_)
        else:
            if self._two_streams:                                       # the sinks wrote p.grad on the per-model (and side) streams
                for s_ in self._streams + self._wgrad_streams:
                    torch.cuda.current_stream().wait_stream(s_)
            for b in self.buckets:
                b.finish()
            for o in self.opts:
                o.step()
        self.iter += 1
        with torch.no_grad():
            miou, _ = miou_device(confusion_matrix_device(ps1, l_target, cfg.num_classes))
        self._join()
        return {"loss": loss.detach(), "sup_loss_1": sup_1.detach(), "sup_loss_2": sup_2.detach(), "cps_loss": cps.detach(),
                "commitment_loss": com_sum.detach(), "prototype_loss": prototype.detach(), "miou": miou,
                "lr": torch.tensor(lr)}

    def sync_buffers(self):
        """Data parallel: BatchNorm running statistics are per-rank (each rank normalises with its own batch statistics -- the
        reference's single-device semantics per rank; parameters and codebooks ARE identical on all ranks).  This broadcasts
        rank 0's buffers to every rank -- an EXPLICIT act (end of training, before an all-rank evaluation).  Nothing on the training
        path calls it: the eval-mode pseudo-label forwards read the running statistics, so a hidden sync (round 2 had one inside
        save_checkpoint) would make the trajectory of ranks > 0 depend on how often checkpoints are written."""
        if vdist.collectives_on():
            for m in self.models:
                for b in m.buffers():
                    dist.broadcast(b.data, src=0)

    def state_dicts(self, sync: bool = False):
        """(model_1, model_2, optimizer_1, optimizer_2) state_dicts for utils.ckpoints.save_ckpoints -- THIS rank's state.
        `sync=True` first broadcasts rank 0's BatchNorm buffers (a collective that changes ranks > 0: see sync_buffers)."""
        if sync:
            self.sync_buffers()
        return (self.models[0].state_dict(), self.models[1].state_dict(), self.opts[0].state_dict(), self.opts[1].state_dict())

    def save_checkpoint(self, filepath, epoch: int = 0, batch_idx: int = 0):
        """The reference's checkpoint dictionary (utils/ckpoints.py:7-13: model_1, model_2, epoch, batch_idx, optimizer_1,
        optimizer_2) plus what a bit-exact resume needs and the reference does not store: the `initted` flags and
        the iteration counter that positions the cosine schedule.  Rank 0 writes ITS state (parameters, codebooks and optimiser
        moments are identical on all ranks; the BatchNorm running statistics in the file are rank 0's); no collective, no rank's
        state is touched -- writing a checkpoint cannot alter the run."""
        extra = {"iter": self.iter}
        if vdist.collectives_on() and vdist.world_size() > 1:
            # BatchNorm running statistics are per-rank: COPIES of every rank's buffers are gathered to rank 0 (no rank's state is
            # touched) and stored beside the reference's keys, so that a data-parallel resume restores each rank's own statistics
            for i, m in enumerate(self.models):
                names = [k for k, _ in m.named_buffers()]
                flat = torch.cat([b.detach().double().reshape(-1) for _, b in m.named_buffers()]) if names else torch.zeros(0, dtype=torch.float64, device=self.device)
                if dist.get_backend() == "gloo":
                    flat = flat.cpu()
                gathered = [torch.zeros_like(flat) for _ in range(vdist.world_size())]
                dist.all_gather(gathered, flat)
                extra[f"bn_buffers_model_{i + 1}"] = {"names": names, "shapes": [tuple(b.shape) for _, b in m.named_buffers()],
                                                       "per_rank": [g.cpu() for g in gathered]}
        if vdist.rank() == 0:
            m1, m2, o1, o2 = self.state_dicts()
            save_ckpoints(m1, m2, epoch, batch_idx, o1, o2, filepath, models=self.models, extra=extra)

    def load_checkpoint(self, filepath):
        """Resume from save_checkpoint's file (or a reference-written one: then the flags and the counter stay as they are).
        Returns (epoch, batch_idx)."""
        state = load_training_state(filepath, map_location=self.device)
        for m, o, i in zip(self.models, self.opts, (1, 2)):
            m.load_state_dict(state[f"model_{i}"])
            o.load_state_dict(state[f"optimizer_{i}"])
            nnf.invalidate_weight_caches(m)
        for m, flags in zip(self.models, state.get("initted") or (None, None)):
            restore_initted(m, flags)
        for i, m in enumerate(self.models):                # data-parallel resume: this rank's own BatchNorm statistics (see save_checkpoint)
            rec = state.get(f"bn_buffers_model_{i + 1}")
            if rec and vdist.rank() < len(rec["per_rank"]):
                flat, off = rec["per_rank"][vdist.rank()], 0
                bufs = dict(m.named_buffers())
                with torch.no_grad():
                    for name, shape in zip(rec["names"], rec["shapes"]):
                        n = 1
                        for d in shape:
                            n *= d
                        if name in bufs:
                            bufs[name].copy_(flat[off:off + n].view(shape).to(bufs[name].dtype))
                        off += n
        self.iter = int(state.get("iter", self.iter))
        return state.get("epoch", 0), state.get("batch_idx", 0)

    def _ce_dice(self, pred, target):
        """0.5 * CE(ignore 255) + criterion (train_vqreptunet1x1v2.py:165-187); with the Dice criterion both terms come from
        one fused pass over the logits."""
        if isinstance(self.criterion, DiceLoss):
            return ce_dice_loss(pred, target, self.criterion.num_classes, 0.5, self.criterion.weight, self.criterion.ignore_index)
        return 0.5 * self.ce(pred, target) + self.criterion(pred, target)

    def supervised_step(self, l_input, l_target) -> torch.Tensor:
        """Plain supervised step of model 1 (Dice + 0.5 CE), for the single-model throughput figure."""
        m = self.models[0]
        self.buckets[0].zero()
        m.train()
        kw = dict(percent=80.0) if self.cfg.recipe == "v1" else dict(th=self.cfg.confidence_threshold)
        pred, closs, _u, ploss = self._fwd(m, l_input, l_target, **kw)
        self._join()
        pred = pred.float()
        loss = self._ce_dice(pred, l_target) + closs.sum() + 0.01 * ploss.float()
        loss.backward()
        nnf.flush_pending_wgrads()
        if self._two_streams:
            torch.cuda.current_stream().wait_stream(self._streams[0])
        self.buckets[0].finish()
        self.opts[0].step()
        return loss.detach()
