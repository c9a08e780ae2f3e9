#!/usr/bin/env python
"""Headline benchmark: train images/sec @512x512 vqreptunet1x1 K=512 (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
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
SIZE, K_CODES = 512, 512


def model_cfg():
    return {"name": "vqreptunet1x1", "params": {
        "encoder_name": "resnet50", "num_classes": 3, "depth": 5,
        "vq_cfg": {"num_embeddings": [0, 0, K_CODES, K_CODES, K_CODES], "distance": "euclidean", "kmeans_init": True},
        "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}


def cpu_baseline():
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
    ref = cps_ref.CPSReference([synth.synth_state_dict(lay, 77), synth.synth_state_dict(lay, 78)], num_embeddings=(0, 0, K_CODES, K_CODES, K_CODES))
    NB, REPS = 1, 3
    l_in, l_tg = synth.blob_images(1, NB, SIZE, cell=32)
    ul_in, _ = synth.blob_images(2, NB, SIZE, cell=32)
    ref.step(l_in, l_tg, ul_in)                         # warm-up (thread pool, primitive caches)
    times = []
    for _ in range(REPS):
        t0 = time.time()
        out = ref.step(l_in, l_tg, ul_in)
        times.append(time.time() - t0)
        print(f"[bench] cpu baseline: CPS iteration on {NB}+{NB} images {times[-1]:.2f}s (loss {out['loss']:.4f})", file=sys.stderr, flush=True)
    dt = sum(times) / len(times)
    return {"value": round(2.0 * NB / dt, 5), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/cps_ref.py::CPSReference.step (the whole CPS iteration, v1 recipe, K = {K_CODES}) on {NB} labelled + {NB} unlabelled "
                      f"image of {SIZE}x{SIZE}, fp32, {cores} threads: mean of {REPS} iterations after a warm-up = {dt:.2f}s per {2 * NB} images"}


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
        sync = torch.zeros(L.vqseg_bn_sync_ints(C), dtype=torch.int32, device=device)
        dg, gy = torch.empty(2, C, device=device), torch.empty_like(ys[0])
        gres = torch.empty_like(ys[0]) if res else None

        def apply(i):
            k = i % sets
            rc = L.vqseg_bn_apply_f(1, ys[k].data_ptr(), rs[k].data_ptr() if res else None, sc.data_ptr(), sh.data_ptr(), M, C, 1, outs[k].data_ptr(), st)
            assert rc == 0, L.vqseg_last_error()

        def bwd(i):
            k = i % sets
            rc = L.vqseg_bn_backward_f(1, gs[k].data_ptr(), outs[k].data_ptr() if res else None, ys[k].data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                       gamma.data_ptr(), sc.data_ptr(), sh.data_ptr(), M, C, 1, 1, 0, ws.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(),
                                       gy.data_ptr(), gres.data_ptr() if res else None, sync.data_ptr(), st)
            assert rc == 0, L.vqseg_last_error()
        for name, fn, nbytes in (("apply", apply, M * C * 2 * (3 if res else 2)), ("backward", bwd, M * C * 2 * (8 if res else 5))):
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
        del ys, gs, rs, outs, gy, gres
    achieved = tot_bytes / tot_us / 1e3
    return {"kernel": "bn_apply_kernel / bn_bwd_reduce_kernel + bn_bwd_apply_kernel (bf16)", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "per_shape": per,
            "note": "algorithmic bytes (apply: y in [+ residual] + out; backward: (g, y) read by the reduce, (g, y) read + g_y [+ g_res] written "
                    "by the apply) / in-stream event time, kernels alone on the chip, HBM-cold operands; the guide's achievable streaming rate is "
                    "~6.3 TB/s of the 8 TB/s peak"}


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
    for t in (8, 4, 2, 1):                                             # vq_group_tiles (csrc/vq_kernels.hip)
        if any(((k + 31) // 32) % t for _n, _c, k in shapes):
            continue
        wgs = sum(((n + 127) // 128 + 7) // 8 * 8 * (((k + 31) // 32) // t) for n, _c, k in shapes)
        if wgs >= 512 or t == 1:
            out = {}
            for rows in ("bf16", "f32"):
                e = table.get(f"WG{wgs}_T{t}_{rows}")
                if e:
                    out[rows] = round(e["hbm_bytes"])
            return out, os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__)))
    return {}, None


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
        have = torch.cuda.device_count()                       # device_count() does not create a HIP context
        if have < n:
            print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); set VQSEG_DIST_REHEARSAL=1 for the one-GPU gloo rehearsal "
                  f"of the N > 1 path", file=sys.stderr)
            return 2
    with socket.socket() as s:
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
    ap.add_argument("--batch", type=int, default=32, help="labelled images per GPU per step (+ as many unlabelled)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="timed region only: skip the all-bf16 step, the supervised step and the roofline_conv steps (profiler passes)")
    ap.add_argument("--eval-amp", action="store_true",
                    help="run the two no-grad pseudo-label forwards under autocast too (all-bf16 step; NOT the reference's "
                         "precision: its trainers run them in fp32, outside autocast)")
    args = ap.parse_args()

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
    cfg = CPSConfig(model=model_cfg(), recipe="v1", total_iters=args.steps + args.warmup + 1,
                    amp_dtype=torch.bfloat16 if args.dtype == "bf16" else None, eval_amp=args.eval_amp)
    trainer = CPSTrainer(cfg, device)
    data = SyntheticCropWeed(SIZE, args.batch, device, seed=42)
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
    if world > 1 and rehearsal:
        # data-parallel invariant: after the same averaged updates every rank holds bit-identical parameters
        chk = torch.stack([p.detach().double().sum() for m in trainer.models for p in m.parameters()]).cpu()
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        if not all(torch.equal(g, gathered[0]) for g in gathered):
            names = [f"model{i}.{k}" for i, m in enumerate(trainer.models) for k, _ in m.named_parameters()]
            bad = [names[j] for j in range(len(names)) if any(g[j] != gathered[0][j] for g in gathered)]
            raise AssertionError(f"ranks diverged in {len(bad)} of {len(names)} parameters, e.g. {bad[:6]} ... {bad[-3:]}")
        trainer.sync_buffers()                                # BatchNorm statistics are per-rank by design; after the sync they agree too
        chk = torch.stack([b.detach().double().sum() for m in trainer.models for b in m.buffers()]).cpu()
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "buffers differ after sync_buffers()"
        in_bwd = [sum(b.launched_in_backward) for b in trainer.buckets]
        note(f"rank {rank}: parameter and (synced) buffer checksums identical on all {world} ranks; buckets reduced inside backward: "
             f"{in_bwd} of {[len(b.buckets) for b in trainer.buckets]}")

    recs = _hip.profile_collect(64 * args.steps) if rank == 0 else []
    # next to the CPS figure, plain forward + backward + Adam of ONE network on the B labelled images
    # (ordinary supervised training throughput) -- measured after, and outside of, the timed region
    (l_in, l_tg), _ul = batches[0]
    sup_s = None
    if not args.no_extras:
        trainer.supervised_step(l_in, l_tg)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        ts = time.perf_counter()
        for _ in range(3):
            trainer.supervised_step(l_in, l_tg)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        tsup = torch.tensor([(time.perf_counter() - ts) / 3], device=device, dtype=torch.float64)
        if multi:
            dist.all_reduce(tsup, op=dist.ReduceOp.MAX)
        sup_s = float(tsup.item())

    # the all-bf16 variant of the step (pseudo-label forwards under autocast as well), reported NEXT to the headline: a build-side
    # speed mode, not the reference's precision -- `value` above is the step whose every forward has the reference's precision
    all_bf16_s = None
    if args.dtype == "bf16" and not args.eval_amp and not args.no_extras:
        trainer.cfg.eval_amp = True
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
        one(0)
        torch.cuda.synchronize()
        if rank == 0:
            _hip.conv_profile_begin(4096)
        for i in range(2):
            one(i)
        torch.cuda.synchronize()
        if rank == 0:
            conv_recs = _hip.conv_profile_collect(4096)
        trainer._two_streams = was_two
    if multi:
        dist.barrier()

    if rank == 0:
        flops = sum(2.0 * n * c * k for n, c, k, _ in recs)
        ms = sum(r[3] for r in recs)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        group = []                                               # the levels of one (grouped) launch: records until a shape repeats
        for n, c, k, _ in recs:
            if (n, c, k) in group:
                break
            group.append((n, c, k))
        n_launch = len(recs) // max(len(group), 1)
        # Bytes the TIMED kernel moves, per launch and row type: the pixel rows in (C * s) + one 8-byte key per row out (the quantised
        # rows are written by the gather kernel, not by this one) -- the figure `traffic` (PMC bytes of the same kernel) compares with.
        # Per step the kernel runs 4 x on bf16 rows (training forwards) and 2 x on fp32 rows (pseudo-label forwards; all bf16 with --eval-amp).
        n_bf16, n_f32 = ((6, 0) if args.eval_amp else (4, 2)) if args.dtype == "bf16" else (0, 6)
        alg = {"bf16": round(sum(n * (c * 2.0 + 8) for n, c, k in group)), "f32": round(sum(n * (c * 4.0 + 8) for n, c, k in group))}
        alg_avg = (n_bf16 * alg["bf16"] + n_f32 * alg["f32"]) / 6.0
        hbm_gbs = alg_avg * n_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr, traffic_src = pmc_traffic(group)
        traffic = round((n_bf16 * tr["bf16"] + n_f32 * tr["f32"]) / 6.0) if ("bf16" in tr and "f32" in tr) else (tr.get("bf16") if n_f32 == 0 else None)
        per_shape = vq_per_level(device, group, bf16_rows=args.dtype == "bf16") if not args.no_extras else {}
        images = 2 * args.batch * world * args.steps
        line = {
            "metric": "train images/sec @512x512 vqreptunet1x1 K=512",
            "value": round(images / elapsed, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "CPS training iteration (2 x vqreptunet1x1, ResNet-50 encoder, K=[0,0,512,512,512], "
                                   "v1 recipe: 6 forwards + 4 backwards + 2 Adam steps) on 512x512x3 images",
                       "images_per_step_per_gpu": 2 * args.batch, "labelled_per_gpu": args.batch,
                       "unlabelled_per_gpu": args.batch, "parallelism": f"dp{world}",
                       "collectives": ("none (one process)" if not multi else "gloo rehearsal, every rank on cuda:0 (timings meaningless)" if rehearsal
                                       else "RCCL" + (" (one-rank drive of the N > 1 code path, VQSEG_DIST_SINGLE)" if world == 1 else "")),
                       "vq_dtype": "f32 (exact fp32 MFMA, bit-exact argmin)", "conv_dtype": args.dtype,
                       "precision_per_forward": {
                           "4 training forwards + 4 backwards": args.dtype + (" (under autocast, like the reference's AMP region; its fp16 -> bf16)" if args.dtype == "bf16" else ""),
                           "2 no-grad pseudo-label forwards": ("bf16 (--eval-amp: NOT the reference's precision)" if (args.eval_amp and args.dtype == "bf16")
                                                               else "fp32 (outside autocast, train_vqreptunet1x1v2.py:143-149)")},
                       "final_loss": round(loss, 5)},
            "roofline": {"kernel": "vq_assign_f32_kernel", "bound": "mfma", "achieved": round(achieved, 2),
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_by_row_type": tr, "algorithmic_bytes_per_launch": round(alg_avg), "algorithmic_bytes_by_row_type": alg,
                         "launch_mix_per_step": {"bf16 rows": n_bf16, "f32 rows": n_f32},
                         "other_roof": {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic bytes of this kernel (rows in + one 8-byte key per row) over the same launch "
                                                "times: the fp32 distance contraction sits far on the MFMA side of the ridge"},
                         "launches": n_launch, "levels_per_launch": len(group), "avg_launch_us": round(ms / max(n_launch, 1) * 1e3, 2),
                         "at_held_clock": {"clock_mhz": 2142, "peak": round(FP32_MFMA_PEAK_TFLOPS * 2142 / 2400, 1),
                                           "frac": round(achieved / (FP32_MFMA_PEAK_TFLOPS * 2142 / 2400), 4),
                                           "source": "profiles/r03_vq_wg_timeline.md: s_memtime / s_memrealtime inside this kernel = 2.14 GHz "
                                                     "(the 157.3 TF/s peak is quoted at 2.4 GHz); `frac` above stays against the nominal peak"},
                         "per_shape": per_shape,
                         "note": "algorithmic flops 2*N*K*C per launch / HIP-event time on the launch stream, all launches inside the "
                                 "timed region; peak = fp32 MFMA (MI355X_MICROARCH.md); the three levels of a forward share ONE launch "
                                 "(longest workgroups first); per_shape = each level ALONE in its own launch, after the timed region"},
        }
        if sup_s is not None:
            line["supervised_step"] = {"images_per_sec": round(args.batch * world / sup_s, 2), "ms_per_step": round(sup_s * 1e3, 2),
                                       "what": "forward + backward + Adam of ONE network on the labelled half of the batch "
                                               "(0.5 CE + Dice + commitment + prototype loss), 3 steps after the timed region"}
        if conv_recs:
            def rate(sel):
                fl = sum(f for f, kd, m in conv_recs if sel(kd))
                ms_ = sum(m for f, kd, m in conv_recs if sel(kd))
                n_ = sum(1 for f, kd, m in conv_recs if sel(kd))
                return {"launches_per_step": n_ // 2, "tflop_per_step": round(fl / 2 / 1e12, 3), "ms_per_step": round(ms_ / 2, 3),
                        "tflops": round(fl / ms_ / 1e9, 1) if ms_ > 0 else 0.0}
            k3 = rate(lambda kd: kd == 300)
            line["roofline_conv"] = {
                "kernel": "conv3x3_patch_kernel / conv_igemm_glds_kernel (every 3x3 bf16 launch: forward + data gradient)",
                "bound": "mfma", "achieved": k3["tflops"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(k3["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "traffic_source": "profiles/r02_conv_patch_pmc.md (per-layer PMC passes of the patch kernel: HBM-side bytes 1.2-4.0x algorithmic, Infinity-Cache hits included; no single per-launch figure exists for an aggregate over 260 launches of 40 shapes)",
                "by_kind": {"3x3 bf16": k3, "1x1 bf16": rate(lambda kd: kd == 100), "3x3 split-3 (fp32-precision eval)": rate(lambda kd: kd == 302),
                            "1x1 split-3": rate(lambda kd: kd == 102), "precise (fp32 activations)": rate(lambda kd: kd % 100 == 1),
                            "3x3 weight gradient bf16": rate(lambda kd: kd == 350), "1x1 weight gradient bf16": rate(lambda kd: kd == 150),
                            "7x7 stem weight gradient": rate(lambda kd: kd // 100 == 7 and kd % 100 >= 50)},
                "whole_step_delivered_tflops": round(sum(f for f, kd, _m in conv_recs if kd % 100 < 50) / 2 / (elapsed / args.steps) / 1e12, 1),
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
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
