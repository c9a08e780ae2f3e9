#!/usr/bin/env python
"""Headline benchmark: train images/sec @512x512 vqreptunet1x1 K=512 (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload cfg2|cfg3|cfg4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launch contract: one process per GPU.  Under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) this process IS one rank.  A plain `python bench.py --gpus N` with N > 1 and no WORLD_SIZE is the launcher: before any
HIP call it starts N fresh rank processes of itself with those variables set (127.0.0.1 rendezvous), relays rank 0's JSON line and
exits with the worst child's return code -- it never touches the GPU and never re-execs.

A step = ONE cross-pseudo-supervision training iteration of the two VQ-UNets on B labelled +
B unlabelled synthetic 512x512 images per GPU (6 forwards, 4 backwards, 2 Adam steps):
"images" = the 2*B input images a step consumes per GPU.  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- the hand-written distance+argmin kernel (vq_assign_f32_kernel): algorithmic flops
                  2*N*K*C of every launch inside the timed region / its HIP-event time, vs the fp32
                  MFMA peak (157.3 TF/s, MI355X_MICROARCH.md)
  cpu_baseline -- the CPU oracle's restatement of the same iteration (oracle/cps_ref.py::CPSReference.step) timed on this
                  host's cores on a bounded sample (1 labelled + 1 unlabelled 512x512 image per iteration, 3 timed iterations);
                  a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# No convolution of the path goes through MIOpen (all are hand-written HIP); should an ATen fallback ever reach it (the plain
# `unet` plumbing model's 3x3 head does), it must not run an exhaustive per-shape search on a fresh box (minutes of silence).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
SIZE, K_CODES = 512, 512           # the metric's configuration (BASELINE.json `metric`; configs[2], "cfg3")

# BASELINE.json `configs` that fit one GPU.  The driver's default run is cfg3 (the configuration the metric is quoted on); cfg2 / cfg4
# are run by hand (profiles/r04_bench_cfg{2,4}.json).  "bs = 64 / GPU" of configs[4] is taken as 64 input images per GPU and step =
# 32 labelled + 32 unlabelled (the reference's `batch_size` is per loader, config/vqreptunet1x1.json:34, i.e. 64 would mean 64 + 64:
# `--batch 64` runs that reading).
WORKLOADS = {
    "cfg2": dict(model="vqreptunet1x1", recipe="v1", size=512, k=256, batch=32, margin=0.0, scale=1.0,
                 what="BASELINE configs[1]: vqreptunet1x1 CWFID 512x512, codebook K = 256, bf16"),
    "cfg3": dict(model="vqreptunet1x1", recipe="v1", size=512, k=512, batch=32, margin=0.0, scale=1.0,
                 what="BASELINE configs[2] (the metric's configuration): vqreptunet1x1 512x512, K = 512"),
    "cfg4": dict(model="vqreptunet1x1v2", recipe="v2", size=1024, k=1024, batch=8, margin=0.5, scale=30.0,
                 what="BASELINE configs[3]: vqreptunet1x1v2 rice_s_n_w 1024x1024, K = 1024, bf16 (v2 recipe: score-mask CPS, CE + Dice)"),
}


def model_cfg(workload: str = "cfg3"):
    w = WORKLOADS[workload]
    return {"name": w["model"], "params": {
        "encoder_name": "resnet50", "num_classes": 3, "depth": 5,
        "vq_cfg": {"num_embeddings": [0, 0, w["k"], w["k"], w["k"]], "distance": "euclidean", "kmeans_init": True},
        "margin": w["margin"], "scale": w["scale"], "use_feature": False, "encoder_weights": None}}


def cpu_baseline(size: int = SIZE, k_codes: int = K_CODES):
    """The CPU oracle's CPS iteration itself (oracle/cps_ref.py::CPSReference.step: 2 eval forwards, 4 training forwards, one backward
    through both networks, 2 Adam steps, the loss / pseudo-label block -- the reference trainer's loop body restated on torch CPU ops,
    pinned to the reference by tests/golden/cps_iter_v1.npz) on a BOUNDED sample of the same workload: 1 labelled + 1 unlabelled
    512x512 image per iteration, fp32, the box's CPU share; one warm-up iteration, then REPS timed ones."""
    from oracle import cps_ref
    from tests import golden_io, synth
    # threads: the CPUs this process may run on, capped at the GPU box's per-GPU CPU share (16) -- asking for all 256
    # hardware threads of the host over-subscribes that share and is ~10x slower
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    lay = golden_io.layout("vqreptunet1x1")
    ref = cps_ref.CPSReference([synth.synth_state_dict(lay, 77), synth.synth_state_dict(lay, 78)], num_embeddings=(0, 0, k_codes, k_codes, k_codes))
    NB, REPS = 1, 3
    l_in, l_tg = synth.blob_images(1, NB, size, cell=32)
    ul_in, _ = synth.blob_images(2, NB, size, cell=32)
    ref.step(l_in, l_tg, ul_in)                         # warm-up (thread pool, primitive caches)
    times = []
    for _ in range(REPS):
        t0 = time.time()
        out = ref.step(l_in, l_tg, ul_in)
        times.append(time.time() - t0)
        print(f"[bench] cpu baseline: CPS iteration on {NB}+{NB} images {times[-1]:.2f}s (loss {out['loss']:.4f})", file=sys.stderr, flush=True)
    dt = sum(times) / len(times)
    return {"value": round(2.0 * NB / dt, 5), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/cps_ref.py::CPSReference.step (the whole CPS iteration, v1 recipe, K = {k_codes}) on {NB} labelled + {NB} unlabelled "
                      f"image of {size}x{size}, fp32, {cores} threads: mean of {REPS} iterations after a warm-up = {dt:.2f}s per {2 * NB} images"}


def bn_roofline(device):
    """HBM-bound side of the step: the BatchNorm passes (apply; backward = reduce + apply) of three of the step's layer shapes, bf16,
    operands rotated through buffer sets larger than the Infinity Cache, in-stream events -- algorithmic bytes / time against the
    HBM peak.  Outside the timed region."""
    from vq_seg_amd import _hip
    L = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    tot_bytes = tot_us = 0.0
    per = {}
    for M, C, res in ((32 * 256 * 256, 64, 0), (32 * 128 * 128, 256, 1), (32 * 32 * 32, 1024, 1)):
        sets = max(2, min(8, int(1.2e9 / (M * C * 2 * 3))))
        sc, sh = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.1
        ys = [torch.randn(M, C, device=device).bfloat16() for _ in range(sets)]
        gs = [torch.randn(M, C, device=device).bfloat16() for _ in range(sets)]
        rs = [torch.randn(M, C, device=device).bfloat16() for _ in range(sets)] if res else None
        outs = [torch.empty_like(t) for t in ys]
        mean, inv, gamma = torch.zeros(C, device=device), torch.ones(C, device=device), torch.ones(C, device=device)
        ws = torch.empty(L.vqseg_bn_backward_workspace_floats(M, C), device=device)
        # the step's default is the two-launch backward (sync = NULL); the one-launch form only when VQSEG_OPTS=py_bn_fused=1 selects it
        from vq_seg_amd import nnf as _nnf
        sync = torch.zeros(L.vqseg_bn_sync_ints(C), dtype=torch.int32, device=device) if _nnf.py_opt("py_bn_fused", 0) else None
        dg, gy = torch.empty(2, C, device=device), torch.empty_like(ys[0])
        # residual layers as the step runs an identity Bottleneck's bn3: the ReLU mask travels as a bit field, the masked shortcut
        # gradient is not stored (vqseg_bn_apply_bits_f / vqseg_bn_backward_bits_f with g_res = NULL)
        bits = [torch.empty(M * C // 8, dtype=torch.uint8, device=device) for _ in range(sets)] if res else None

        def apply(i):
            k = i % sets
            if res:
                rc = L.vqseg_bn_apply_bits_f(ys[k].data_ptr(), rs[k].data_ptr(), sc.data_ptr(), sh.data_ptr(), M, C, outs[k].data_ptr(),
                                             bits[k].data_ptr(), st)
            else:
                rc = L.vqseg_bn_apply_f(1, ys[k].data_ptr(), None, sc.data_ptr(), sh.data_ptr(), M, C, 1, outs[k].data_ptr(), st)
            assert rc == 0, L.vqseg_last_error()

        def bwd(i):
            k = i % sets
            if res:
                rc = L.vqseg_bn_backward_bits_f(gs[k].data_ptr(), bits[k].data_ptr(), ys[k].data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                gamma.data_ptr(), M, C, 1, 0, ws.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), gy.data_ptr(), None,
                                                sync.data_ptr() if sync is not None else None, st)
            else:
                rc = L.vqseg_bn_backward_f(1, gs[k].data_ptr(), None, ys[k].data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                           gamma.data_ptr(), sc.data_ptr(), sh.data_ptr(), M, C, 1, 1, 0, ws.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(),
                                           gy.data_ptr(), None, sync.data_ptr() if sync is not None else None, st)
            assert rc == 0, L.vqseg_last_error()
        if res:
            for k in range(sets):                                           # the backward legs read bits the apply legs wrote
                apply(k)
        for name, fn, nbytes in (("apply", apply, M * C * 2 * (3 if res else 2) + (M * C // 8 if res else 0)),
                                 ("backward", bwd, M * C * 2 * 5 + (2 * (M * C // 8) if res else 0))):
            fn(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8):
                fn(i + 1)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 8 * 1e3
            tot_bytes += nbytes
            tot_us += us
            per[f"{name} M{M}xC{C}{'+res' if res else ''}"] = {"us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1)}
        del ys, gs, rs, outs, gy, bits
    achieved = tot_bytes / tot_us / 1e3
    return {"kernel": "bn_apply_kernel / bn_bwd_reduce_kernel + bn_bwd_apply_kernel (bf16)", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "per_shape": per,
            "note": "algorithmic bytes (apply: y in [+ residual] + out [+ the ReLU mask bits of a residual layer, 1 bit per element]; "
                    "backward: (g, y [, bits]) read by the reduce, (g, y [, bits]) read + g_y written by the apply -- r4: a residual layer's "
                    "masked shortcut gradient is no longer stored, 5 1/8 tensor passes instead of r3's 8) / in-stream event time, kernels "
                    "alone on the chip, HBM-cold operands; the guide's achievable streaming rate is ~6.3 TB/s of the 8 TB/s peak"}


def pmc_traffic(shapes):
    """HBM bytes per launch of the distance+argmin kernel from the committed rocprofv3 --pmc summary of this same command
    (profiles/*_vq_assign_pmc.json: FETCH_SIZE x 2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes), looked up by the
    launch's grid and row type: one launch serves the levels `shapes` = [(N, C, K), ...] of a forward; -> ({"bf16": bytes, "f32":
    bytes}, source file).  PMC passes serialise kernels, so they cannot run inside the timed bench."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_vq_assign_pmc.json")))
    if not files:
        return {}, None
    table = json.load(open(files[-1]))["shapes"]
    def plain_wgs(t):                                                  # vq_group_wgs (csrc/vq_kernels.hip): unpadded row blocks
        return sum((n + 127) // 128 * (((k + 31) // 32) // t) for n, _c, k in shapes)
    divides = lambda t: not any(((k + 31) // 32) % t for _n, _c, k in shapes)       # noqa: E731
    fine = next((t for t in (8, 4) if divides(t) and plain_wgs(t) >= 4096), None)   # vq_group_tiles: r4's finer split first
    for t in ((fine,) if fine else ()) + (8, 4, 2, 1):
        if not divides(t):
            continue
        wgs = sum(((n + 127) // 128 + 7) // 8 * 8 * (((k + 31) // 32) // t) for n, _c, k in shapes)   # the launched grid (XCD-padded)
        if t == fine or plain_wgs(t) >= 512 or t == 1:
            out = {}
            for rows in ("bf16", "f32"):
                e = table.get(f"WG{wgs}_T{t}_{rows}")
                if e:
                    out[rows] = round(e["hbm_bytes"])
            return out, os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__)))
    return {}, None


def pmc_traffic_filter(shapes):
    """HBM bytes per grouped launch of the bf16 candidate filter's three kernels from the same committed summary (its "bf16_filter"
    table, keyed by kernel and grid) -> (bytes or None, {kernel: bytes}, source)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_vq_assign_pmc.json")))
    if not files:
        return None, {}, None
    table = json.load(open(files[-1])).get("bf16_filter", {})
    fend = sum(((n + 127) // 128 + 7) // 8 * 8 * (k // 256) for n, _c, k in shapes)
    rend = sum((n + 255) // 256 for n, _c, k in shapes)
    send = sum(min(max((n // 256 + 63) // 64, 4), 16) * 64 for n, _c, k in shapes)
    parts = {}
    for kern, wgs in (("vq_filter_bf16_kernel", fend), ("vq_resolve_kernel", rend), ("vq_rescore_kernel", send)):
        e = table.get(f"{kern}_WG{wgs}")
        if e is None:
            return None, {}, None
        parts[kern] = round(e["hbm_bytes"])
    return sum(parts.values()), parts, os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__)))


def vq_per_level(device, group, bf16_rows=True, reps=10):
    """Each level of the grouped launch ALONE (its own launch of the same kernel, in-stream events, outside the timed region): the
    grouped launch's time cannot be split by level, so the per-shape rates are measured, not apportioned."""
    from vq_seg_amd import _hip
    out = {}
    for n, c, k in group:
        x = torch.relu(torch.randn(n, c, device=device))
        x = x.to(torch.bfloat16) if bf16_rows else x
        W = torch.relu(torch.randn(k, c, device=device))
        prep = _hip.vq_prepare(W)
        for _ in range(3):
            _hip.vq_assign(x, W, prepared=prep)
        torch.cuda.synchronize()
        _hip.profile_begin(2 * reps)
        for _ in range(reps):
            _hip.vq_assign(x, W, prepared=prep)
        recs = _hip.profile_collect(2 * reps)
        ms = sorted(r[3] for r in recs)[len(recs) // 2]
        out[f"N{n}xC{c}xK{k}"] = {"alone_us": round(ms * 1e3, 1), "tflops": round(2.0 * n * c * k / ms / 1e9, 1),
                                  "frac": round(2.0 * n * c * k / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4)}
        del x, W, prep
    return out


def measured_vq_clock(device, group):
    """The shader clock INSIDE the distance + argmin kernel, measured in this run: one grouped launch of the
    timeline build of the same kernel (libvqseg_hip_tl.so: -DVQ_TIMELINE=1, s_memtime / s_memrealtime stamps of wave 0 of every
    workgroup) on fp32 rows of the step's shapes, right after the timed region.  -> dict or None (library not built)."""
    import ctypes
    import numpy as np
    from vq_seg_amd import _hip
    path = os.path.join(ROOT, "vq_seg_amd", "libvqseg_hip_tl.so")
    if not os.path.exists(path) or not group:
        return None
    try:
        H = _hip.bind(path)                                     # a stale build (other ABI) must not take the bench down: no clock then
        H.vqseg_debug_timeline.argtypes = [ctypes.c_void_p]
    except (OSError, AttributeError) as exc:
        print(f"[bench] timeline build not usable ({exc}); run __graft_entry__.build()", file=sys.stderr)
        return None
    rows = [torch.relu(torch.randn(n, c, device=device)) for n, c, k in group]
    books = [torch.relu(torch.randn(k, c, device=device)) for n, c, k in group]
    preps = [_hip.vq_prepare(w) for w in books]
    n_wg = sum(((n + 127) // 128 + 7) // 8 * 8 * ((k + 31) // 32) for n, c, k in group)       # upper bound (one 32-code tile per workgroup)
    for _ in range(2):
        _hip.vq_forward_group(rows, books, preps, False, [1.0] * len(group), handle=H)
    torch.cuda.synchronize()
    tl = torch.zeros(n_wg, 16, dtype=torch.int64, device=device)
    H.vqseg_debug_timeline(tl.data_ptr())
    _hip.vq_forward_group(rows, books, preps, False, [1.0] * len(group), handle=H)
    torch.cuda.synchronize()
    H.vqseg_debug_timeline(None)
    t = tl.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    if len(t) == 0:
        return None
    us = (t[:, 10] - t[:, 0]).astype(np.float64) * 0.01                  # s_memrealtime: 100 MHz
    ck = (t[:, 11] - t[:, 1]).astype(np.float64)
    mhz = ck[us > 0] / us[us > 0]
    return {"clock_mhz": round(float(np.median(mhz)), 1), "p5": round(float(np.percentile(mhz, 5)), 1), "p95": round(float(np.percentile(mhz, 95)), 1),
            "workgroups": int(len(mhz))}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: become the launcher.  Starts N fresh children of this same
    script, one per GPU, with the rendezvous variables torch.distributed.run would set; rank 0's stdout (the one JSON line) is this
    process's stdout, every rank's stderr is passed through.  The launcher has not initialised HIP (importing torch does not) and does
    not exec: children are ordinary subprocesses.  Return code = the worst child's; when one rank dies the others are terminated (by
    PID) instead of being left waiting in a collective."""
    import socket
    import subprocess
    rehearsal = os.environ.get("VQSEG_DIST_REHEARSAL") == "1"
    if not rehearsal:
        # the launcher never asks the HIP runtime anything: the GPUs are counted from the KFD topology (sysfs); a node that shows
        # fewer is reported here, anything else by the ranks themselves
        have, topo = 0, "/sys/class/kfd/kfd/topology/nodes"
        for node in (os.listdir(topo) if os.path.isdir(topo) else []):      # no KFD topology = no amdgpu driver = no GPU
            try:
                with open(f"{topo}/{node}/properties") as f:
                    have += any(line.startswith("simd_count") and int(line.split()[1]) > 0 for line in f)
            except (OSError, ValueError, IndexError):
                have += 1                                      # an unreadable node: count it, the ranks will find out
        if have < n:
            print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); set VQSEG_DIST_REHEARSAL=1 for the one-GPU gloo rehearsal "
                  f"of the N > 1 path", file=sys.stderr)
            return 2
    with socket.socket() as s:                                 # (a port another process grabs in between makes rank 0 fail loudly at bind)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                print(f"bench.py launcher: rank {r} exited with {rc}", file=sys.stderr)
                worst = worst or rc
                for o in live:                                   # exact PIDs of our own children
                    procs[o].terminate()
        time.sleep(0.2)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="labelled images per GPU per step (+ as many unlabelled); 0 = the workload's default "
                                                         "(32 at 512x512, 8 at 1024x1024)")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS), help="BASELINE.json configs[1] / [2] / [3]; cfg3 = the metric's configuration")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="timed region only: skip the all-bf16 step, the supervised step and the roofline_conv steps (profiler passes)")
    ap.add_argument("--eval-amp", action="store_true",
                    help="run the two no-grad pseudo-label forwards under autocast too (all-bf16 step; NOT the reference's "
                         "precision: its trainers run them in fp32, outside autocast)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = wl["batch"]
    size, k_codes = wl["size"], wl["k"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # VQSEG_DIST_REHEARSAL=1: every rank on cuda:0 with gloo -- a functional rehearsal of the N > 1 code path on a
    # one-GPU box (gradient sinks, buckets, distributed k-means init); its timings mean nothing.
    rehearsal = os.environ.get("VQSEG_DIST_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # VQSEG_DIST_SINGLE=1 (with the rendezvous variables set, e.g. under torch.distributed.run --nproc-per-node 1): a ONE-rank RCCL
    # process group and the whole N > 1 code path on top of it (vq_seg_amd.dist.collectives_on) -- what a one-GPU box can
    # exercise of the RCCL path: communicator initialisation, ReduceOp.AVG, async handles, stream ordering.
    multi = world > 1 or os.environ.get("VQSEG_DIST_SINGLE") == "1"      # a process group exists: barriers / reductions run
    if multi:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)      # "nccl" IS RCCL on ROCm

    from vq_seg_amd import _hip
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed

    t_start = time.perf_counter()
    cfg = CPSConfig(model=model_cfg(args.workload), recipe=wl["recipe"], total_iters=args.steps + args.warmup + 1,
                    amp_dtype=torch.bfloat16 if args.dtype == "bf16" else None, eval_amp=args.eval_amp)
    trainer = CPSTrainer(cfg, device)
    data = SyntheticCropWeed(size, args.batch, device, seed=42)
    batches = [(data.labelled(), data.unlabelled()) for _ in range(2)]      # resident in HBM before timing
    torch.cuda.synchronize()

    def one(i):
        (l_in, l_tg), ul_in = batches[i % len(batches)]
        return trainer.step(l_in, l_tg, ul_in, epoch_frac=0.0)

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        one(i)
        torch.cuda.synchronize()
        note(f"warm-up step {i + 1}/{args.warmup} done")
    if multi:
        dist.barrier()
    if rank == 0:
        _hip.profile_begin(64 * args.steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one(i)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss = float(out["loss"])
    note(f"{args.steps} timed steps done in {elapsed:.2f}s")
    dp_check = None
    if multi:
        # Self-validation of the N > 1 path in EVERY multi-rank run (RCCL on real GPUs as well as the gloo rehearsal):
        # after the same averaged updates every rank must hold bit-identical parameters -- checksums are all-gathered and compared --
        # and the gradient buckets must have been reduced from inside backward, in the same order on every rank.
        dev_c = device if not rehearsal else torch.device("cpu")
        chk = torch.stack([p.detach().double().sum() for m in trainer.models for p in m.parameters()]).to(dev_c)
        gathered = [torch.zeros_like(chk) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, chk)
        identical = all(torch.equal(g, gathered[0]) for g in gathered)
        seq = torch.tensor([bi for b in trainer.buckets for bi in b.launch_sequence[:64]] + [-1] * 128, device=dev_c)[:128]
        seqs = [torch.zeros_like(seq) for _ in range(dist.get_world_size())]
        dist.all_gather(seqs, seq)
        same_seq = all(torch.equal(q, seqs[0]) for q in seqs)
        dp_check = {"world_size_seen": dist.get_world_size(), "ranks_identical": bool(identical), "same_bucket_sequence_on_all_ranks": bool(same_seq),
                    "buckets_reduced_in_backward": [int(sum(b.launched_in_backward)) for b in trainer.buckets],
                    "buckets": [len(b.buckets) for b in trainer.buckets], "backend": dist.get_backend()}
        if not identical:
            names = [f"model{i}.{k}" for i, m in enumerate(trainer.models) for k, _ in m.named_parameters()]
            bad = [names[j] for j in range(len(names)) if any(g[j] != gathered[0][j] for g in gathered)]
            raise AssertionError(f"ranks diverged in {len(bad)} of {len(names)} parameters, e.g. {bad[:6]} ... {bad[-3:]}")
        assert same_seq, "the ranks issued their bucket all-reduces in different orders"
        if rehearsal and world > 1:
            trainer.sync_buffers()                            # BatchNorm statistics are per-rank by design; after the sync they agree too
            chk = torch.stack([b.detach().double().sum() for m in trainer.models for b in m.buffers()]).cpu()
            gathered = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(gathered, chk)
            assert all(torch.equal(g, gathered[0]) for g in gathered), "buffers differ after sync_buffers()"
        note(f"rank {rank}: parameter checksums identical on all {dist.get_world_size()} ranks; buckets reduced inside backward: "
             f"{dp_check['buckets_reduced_in_backward']} of {dp_check['buckets']}")

    recs_all = _hip.profile_collect(64 * args.steps, with_kind=True) if rank == 0 else []
    recs = [r[:4] for r in recs_all if r[4] != 2]              # the exact fp32-MFMA kernel (fp32 rows; bf16 rows without the filter)
    recs_f = [r[:4] for r in recs_all if r[4] == 2]            # the bf16 candidate filter + exact re-score (bf16 rows)
    # next to the CPS figure, plain forward + backward + Adam of ONE network on the B labelled images
    # (ordinary supervised training throughput) -- measured after, and outside of, the timed region
    (l_in, l_tg), _ul = batches[0]
    sup_s = None
    # (the gloo rehearsal moves every gradient through the host, ~9 s per CPS step, and its timings mean nothing: one repetition of
    # each extra leg and no warm-ups there -- what it checks is that every rank runs every leg)
    n_sup, n_conv = (1, 1) if rehearsal else (3, 2)
    if not args.no_extras:
        if not rehearsal:
            trainer.supervised_step(l_in, l_tg)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        ts = time.perf_counter()
        for _ in range(n_sup):
            trainer.supervised_step(l_in, l_tg)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        tsup = torch.tensor([(time.perf_counter() - ts) / n_sup], device=device, dtype=torch.float64)
        if multi:
            dist.all_reduce(tsup, op=dist.ReduceOp.MAX)
        sup_s = float(tsup.item())

    # the all-bf16 variant of the step (pseudo-label forwards under autocast as well), reported NEXT to the headline: a build-side
    # speed mode, not the reference's precision -- `value` above is the step whose every forward has the reference's precision
    all_bf16_s = None
    if args.dtype == "bf16" and not args.eval_amp and not args.no_extras:
        trainer.cfg.eval_amp = True
        if not rehearsal:
            one(0)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        ta = time.perf_counter()
        for i in range(args.steps):
            one(i)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        tt = torch.tensor([time.perf_counter() - ta], device=device, dtype=torch.float64)
        if multi:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        all_bf16_s = float(tt.item())
        trainer.cfg.eval_amp = False

    # roofline_conv: two more steps of the SAME configuration with every convolution launch bracketed by HIP events, on ONE
    # stream (with two streams a launch's in-stream time would include the other network's kernels sharing the chip).  Outside
    # the timed region: ~700 event pairs per step would perturb the headline.
    conv_recs = []
    if not args.no_extras:
        # EVERY rank runs these steps (they hold the gradient all-reduces: a rank that skipped them would leave the others
        # waiting in a collective); rank 0 alone brackets its launches with events
        was_two = trainer._two_streams
        trainer._two_streams = False
        if not rehearsal:
            one(0)
        torch.cuda.synchronize()
        if rank == 0:
            _hip.conv_profile_begin(4096)
        for i in range(n_conv):
            one(i)
        torch.cuda.synchronize()
        if rank == 0:
            conv_recs = _hip.conv_profile_collect(4096, with_shape=True)
        trainer._two_streams = was_two
    if multi:
        dist.barrier()

    if rank == 0:
        def level_group(rs):                                     # the levels of one (grouped) launch: records until a shape repeats
            grp = []
            for n, c, k, _ in rs:
                if (n, c, k) in grp:
                    break
                grp.append((n, c, k))
            return grp

        group = level_group(recs) or level_group(recs_f)
        images = 2 * args.batch * world * args.steps
        # ---- the exact fp32-MFMA kernel (the pseudo-label forwards' fp32 rows; every launch when the bf16 filter is off)
        roof = None
        if recs:
            flops = sum(2.0 * n * c * k for n, c, k, _ in recs)
            ms = sum(r[3] for r in recs)
            achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            n_launch = len(recs) // max(len(group), 1)
            per_step = n_launch / max(args.steps, 1)
            # Bytes the TIMED kernel moves, per launch and row type: the pixel rows in (C * s) + one 8-byte key per row out (the quantised
            # rows are written by the gather kernel, not by this one) -- the figure `traffic` (PMC bytes of the same kernel) compares with.
            n_f32 = 0 if (args.eval_amp and args.dtype == "bf16") else (2 if args.dtype == "bf16" else 6)
            n_bf16 = round(per_step) - n_f32                         # bf16 rows reach this kernel only with the candidate filter off
            alg = {"bf16": round(sum(n * (c * 2.0 + 8) for n, c, k in group)), "f32": round(sum(n * (c * 4.0 + 8) for n, c, k in group))}
            alg_avg = (n_bf16 * alg["bf16"] + n_f32 * alg["f32"]) / max(n_bf16 + n_f32, 1)
            hbm_gbs = alg_avg * n_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr, traffic_src = pmc_traffic(group)
            traffic = None
            if (n_bf16 == 0 or "bf16" in tr) and (n_f32 == 0 or "f32" in tr):
                traffic = round((n_bf16 * tr.get("bf16", 0) + n_f32 * tr.get("f32", 0)) / max(n_bf16 + n_f32, 1))
            per_shape = vq_per_level(device, group, bf16_rows=False) if not args.no_extras else {}
            clk = measured_vq_clock(device, group) if not args.no_extras else None
            roof = {"kernel": "vq_assign_f32_kernel", "bound": "mfma", "achieved": round(achieved, 2),
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_by_row_type": tr, "algorithmic_bytes_per_launch": round(alg_avg), "algorithmic_bytes_by_row_type": alg,
                    "launch_mix_per_step": {"bf16 rows": n_bf16, "f32 rows": n_f32},
                    "other_roof": {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                                   "note": "algorithmic bytes of this kernel (rows in + one 8-byte key per row) over the same launch "
                                           "times: the fp32 distance contraction sits far on the MFMA side of the ridge"},
                    "launches": n_launch, "levels_per_launch": len(group), "avg_launch_us": round(ms / max(n_launch, 1) * 1e3, 2),
                    "per_shape": per_shape,
                    "note": "algorithmic flops 2*N*K*C per launch / HIP-event time on the launch stream, all launches of this kernel inside "
                            "the timed region (the pseudo-label forwards' fp32 rows; bf16 rows go through the candidate filter: "
                            "roofline_vq_bf16); peak = fp32 MFMA (MI355X_MICROARCH.md); the three levels of a forward share ONE launch "
                            "(longest workgroups first); per_shape = each level ALONE in its own launch, after the timed region"}
            if clk is not None:
                held = FP32_MFMA_PEAK_TFLOPS * clk["clock_mhz"] / 2400.0
                roof["at_held_clock"] = dict(clk, peak=round(held, 1), frac=round(achieved / held, 4),
                                             source="MEASURED in this run: s_memtime / s_memrealtime of every workgroup of one stamped grouped launch of "
                                                    "the timeline build (libvqseg_hip_tl.so) right after the timed region; the 157.3 TF/s peak is quoted "
                                                    "at 2.4 GHz; `frac` above stays against the nominal peak")
        # ---- the bf16 candidate filter + exact re-score (the training forwards' bf16 rows)
        roof_f = None
        if recs_f:
            grp_f = level_group(recs_f)
            ms_f = sum(r[3] for r in recs_f)
            nl_f = len(recs_f) // max(len(grp_f), 1)
            alg_fl = sum(2.0 * n * c * k for n, c, k, _ in recs_f)
            exe_fl = 2.0 * alg_fl                                  # hi + lo parts of the codebook: twice the algorithmic contraction (re-score: < 1 %)
            alg_b = sum(n * (c * 2.0 + 8) for n, c, k in grp_f)
            tf_alg, tf_exe = alg_fl / (ms_f * 1e-3) / 1e12, exe_fl / (ms_f * 1e-3) / 1e12
            gbs = alg_b * nl_f / (ms_f * 1e-3) / 1e9
            tr_f, tr_parts, tr_src = pmc_traffic_filter(grp_f)
            roof_f = {"kernel": "vq_filter_bf16_kernel + vq_resolve_kernel + vq_rescore_kernel (bf16 rows: bf16-MFMA candidate filter, exact "
                                "fmaf-chain re-score of the candidates; indices and distances identical to vq_assign_f32_kernel)",
                      "bound": "mfma", "achieved": round(tf_exe, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(tf_exe / BF16_MFMA_PEAK_TFLOPS, 4),
                      "algorithmic_tflops": round(tf_alg, 1), "executed_over_algorithmic_flops": 2.0,
                      "vs_fp32_mfma_peak": round(tf_alg / FP32_MFMA_PEAK_TFLOPS, 3),
                      "other_roof": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                     "algorithmic_bytes_per_launch": round(alg_b)},
                      "traffic": tr_f, "traffic_by_kernel": tr_parts, "traffic_source": tr_src,
                      "launches": nl_f, "levels_per_launch": len(grp_f), "avg_launch_us": round(ms_f / max(nl_f, 1) * 1e3, 2),
                      "note": "HIP-event time from the filter kernel's launch to the end of the re-score (three kernels), all launches inside the "
                              "timed region; achieved = EXECUTED bf16 flops (2 x 2*N*K*C: the codebook's hi and lo parts) against the dense bf16 "
                              "MFMA peak, algorithmic_tflops = 2*N*K*C / time (what the exact kernel is measured by: SURVEY 8d's per-row figure); "
                              "with bf16 operands and K = 512 the path sits near the HBM / MFMA ridge (SURVEY 8d): both fractions are given"}
            if recs:
                roof_f["speedup_over_exact_kernel_per_launch"] = round((sum(r[3] for r in recs) / max(len(recs), 1)) / (ms_f / max(len(recs_f), 1)), 2)
        if roof is None:                                          # every launch took the filter (--eval-amp): it IS the dominant VQ kernel
            roof, roof_f = roof_f, None
        model_name = wl["model"]
        line = {
            "metric": f"train images/sec @{size}x{size} {model_name} K={k_codes}",
            "value": round(images / elapsed, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"CPS training iteration (2 x {model_name}, ResNet-50 encoder, K=[0,0,{k_codes},{k_codes},{k_codes}], "
                                   f"{wl['recipe']} recipe: 6 forwards + 4 backwards + 2 Adam steps) on {size}x{size}x3 images -- {wl['what']}",
                       "baseline_config": args.workload,
                       "images_per_step_per_gpu": 2 * args.batch, "labelled_per_gpu": args.batch,
                       "unlabelled_per_gpu": args.batch,
                       "batch_reading": "BASELINE's 'bs=64/GPU' is taken as 64 input images per GPU and step = 32 labelled + 32 unlabelled; "
                                        "`--batch 64` runs the other reading (64 + 64: the reference's batch_size is per loader)",
                       "parallelism": f"dp{world}",
                       "collectives": ("none (one process)" if not multi else "gloo rehearsal, every rank on cuda:0 (timings meaningless)" if rehearsal
                                       else "RCCL" + (" (one-rank drive of the N > 1 code path, VQSEG_DIST_SINGLE)" if world == 1 else "")),
                       "vq_dtype": "f32 arithmetic, bit-exact argmin (fp32 rows: exact fp32 MFMA; bf16 rows: bf16-MFMA candidate filter + exact re-score)",
                       "conv_dtype": args.dtype,
                       "precision_per_forward": {
                           "4 training forwards + 4 backwards": args.dtype + (" (under autocast, like the reference's AMP region; its fp16 -> bf16)" if args.dtype == "bf16" else ""),
                           "2 no-grad pseudo-label forwards": ("bf16 (--eval-amp: NOT the reference's precision)" if (args.eval_amp and args.dtype == "bf16")
                                                               else "fp32 (outside autocast, train_vqreptunet1x1v2.py:143-149)")},
                       "final_loss": round(loss, 5)},
            "roofline": roof,
        }
        if roof_f is not None:
            line["roofline_vq_bf16"] = roof_f
        if dp_check is not None:
            line["dp_check"] = dp_check
        if sup_s is not None:
            line["supervised_step"] = {"images_per_sec": round(args.batch * world / sup_s, 2), "ms_per_step": round(sup_s * 1e3, 2),
                                       "what": "forward + backward + Adam of ONE network on the labelled half of the batch "
                                               "(0.5 CE + Dice + commitment + prototype loss), 3 steps after the timed region"}
        if conv_recs:
            conv_shapes = [r[3] for r in conv_recs]
            conv_recs = [(r[0], r[1], r[2]) for r in conv_recs]

            def rate(sel):
                fl = sum(f for f, kd, m in conv_recs if sel(kd))
                ms_ = sum(m for f, kd, m in conv_recs if sel(kd))
                n_ = sum(1 for f, kd, m in conv_recs if sel(kd))
                return {"launches_per_step": n_ // n_conv, "tflop_per_step": round(fl / n_conv / 1e12, 3), "ms_per_step": round(ms_ / n_conv, 3),
                        "tflops": round(fl / ms_ / 1e9, 1) if ms_ > 0 else 0.0}
            k3 = rate(lambda kd: kd == 300)
            # algorithmic bytes of the same launches: input rows + output rows + the weight, bf16 (pixels from the recorded flops)
            alg, n3 = 0.0, 0
            for (f, kd, _m), sh in zip(conv_recs, conv_shapes):
                if kd == 300 and sh[1] > 0 and sh[2] > 0:
                    px = f / (2.0 * 9 * sh[1] * sh[2])
                    st = max(sh[3] // 10, 1)
                    alg += 2.0 * (px * st * st * sh[1] + px * sh[2] + 9 * sh[1] * sh[2])
                    n3 += 1
            conv_traffic, conv_traffic_src = None, "no committed counter summary found"
            pmc_conv = os.path.join(ROOT, "profiles", "r04_conv_step_pmc.json")
            if os.path.exists(pmc_conv):
                with open(pmc_conv) as f_:
                    conv_traffic = int(json.load(f_)["conv3x3_bf16_fwd_dgrad"]["hbm_bytes_per_launch"])
                conv_traffic_src = ("profiles/r04_conv_step_pmc.json (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE passes over this command, tools/pmc_conv_step.sh): average "
                                    "HBM-side bytes per launch over the step's 3x3 bf16 forward / data-gradient launches (Infinity-Cache hits are counted; a launch "
                                    "re-reads its input rows once per 128-channel output chunk, its weights once per pixel tile)")
            line["roofline_conv"] = {
                "kernel": "conv3x3_patch_kernel / conv_igemm_glds_kernel (every 3x3 bf16 launch: forward + data gradient)",
                "bound": "mfma", "achieved": k3["tflops"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(k3["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": conv_traffic,
                "traffic_source": conv_traffic_src,
                "algorithmic_bytes_per_launch": int(alg / n3) if n3 else None,
                "other_roof": {"bound": "hbm", "achieved": round(alg / n_conv / (k3["ms_per_step"] * 1e-3) / 1e9, 1) if k3["ms_per_step"] > 0 else 0.0, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "note": "algorithmic bytes of the same launches over the same event time"},
                "by_kind": {"3x3 bf16": k3, "1x1 bf16": rate(lambda kd: kd == 100), "3x3 split-3 (fp32-precision eval)": rate(lambda kd: kd == 302),
                            "1x1 split-3": rate(lambda kd: kd == 102), "precise (fp32 activations)": rate(lambda kd: kd % 100 == 1),
                            "3x3 weight gradient bf16": rate(lambda kd: kd == 350), "1x1 weight gradient bf16": rate(lambda kd: kd == 150),
                            "7x7 stem weight gradient": rate(lambda kd: kd // 100 == 7 and kd % 100 >= 50)},
                "whole_step_delivered_tflops": round(sum(f for f, kd, _m in conv_recs if kd % 100 < 50) / n_conv / (elapsed / args.steps) / 1e12, 1),
                "note": "algorithmic flops 2*KH*KW*Cin*Cout*pixels per launch / in-stream HIP-event time, two extra steps after the timed "
                        "region with both networks on ONE stream (kernels alone on the chip); split-3 launches are counted at the LOGICAL "
                        "convolution's flops (they execute 3x that on the MFMA pipes); weight-gradient kernels are listed by_kind (kernel only, the slab "
                        "sums that follow are separate launches) and not part of achieved / whole_step_delivered; "
                        "whole_step_delivered = all convolution forward/data-gradient flops of a step / the headline's step time"}
        if not args.no_extras:
            line["roofline_bn"] = bn_roofline(device)
        if all_bf16_s is not None:
            line["all_bf16_step"] = {"images_per_sec": round(images / all_bf16_s, 3), "ms_per_step": round(all_bf16_s / args.steps * 1e3, 3),
                                     "what": "the same step with the two pseudo-label forwards under bf16 autocast too (CPSConfig.eval_amp=True); "
                                             "narrower than the reference's trainer there, so it is NOT the headline"}
        if world == 1 and not args.no_cpu_baseline and wl["recipe"] == "v1":      # (the oracle's CPS iteration restates the v1 loop body)
            line["cpu_baseline"] = cpu_baseline(size, k_codes)
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
