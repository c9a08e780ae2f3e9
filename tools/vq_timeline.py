"""Per-workgroup timeline of the distance + argmin kernel: where the time outside the MFMA main loop goes.

    make -C vq_seg_amd/csrc timeline            # debug build libvqseg_hip_tl.so (-DVQ_TIMELINE=1: clock stamps of wave 0)
    python tools/vq_timeline.py [out.md]        # on the MI355X box

Runs the grouped launch of the bench's three levels (B = 32 at 512^2, K = 512, bf16 rows -> N = 131072 / 32768 / 8192 rows of
512 / 1024 / 2048 channels: 2688 workgroups of 4 waves, 2 per CU) and reads, per workgroup, (s_memrealtime, s_memtime) at: kernel
entry, first MFMA (prologue done), main loop issued, running minima done, before the key atomic, exit -- plus HW_ID / XCC_ID, so
that workgroups can be put back on their CU.  s_memrealtime ticks at a constant 100 MHz; s_memtime / s_memrealtime = the shader
clock the kernel actually ran at."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VQSEG_LIB"] = os.path.join(ROOT, "vq_seg_amd", "libvqseg_hip_tl.so")
import torch  # noqa: E402
from vq_seg_amd import _hip  # noqa: E402

assert _hip.LIB_PATH.endswith("_tl.so"), _hip.LIB_PATH
dev = torch.device("cuda:0")
L = _hip.lib()
L.vqseg_debug_timeline.argtypes = [ctypes.c_void_p]
shapes = ((131072, 512, 512), (32768, 1024, 512), (8192, 2048, 512))
rows = [torch.relu(torch.randn(n, c, device=dev)).to(torch.bfloat16) for n, c, k in shapes]
books = [torch.relu(torch.randn(k, c, device=dev)) for n, c, k in shapes]
preps = [_hip.vq_prepare(w) for w in books]
n_wg = sum((n // 128 + 7) // 8 * 8 * (k // 256) for n, c, k in shapes)
for _ in range(3):
    _hip.vq_forward_group(rows, books, preps, False, [1.0] * 3)
torch.cuda.synchronize()
tl = torch.zeros(n_wg, 16, dtype=torch.int64, device=dev)
L.vqseg_debug_timeline(tl.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L.vqseg_profile_begin(16)
_hip.vq_forward_group(rows, books, preps, False, [1.0] * 3)
recs = _hip.profile_collect(16)
torch.cuda.synchronize()
L.vqseg_debug_timeline(None)
t = tl.cpu().numpy().astype(np.int64)
launch_us = sum(r[3] for r in recs) * 1e3
ok = t[:, 0] != 0
t = t[ok]
rt = t[:, 0:12:2].astype(np.float64) * 0.01          # microseconds (100 MHz)
ck = t[:, 1:12:2].astype(np.float64)
hw, xcc, lvl = t[:, 12], t[:, 13] & 0xF, t[:, 14]
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
t0 = rt[:, 0].min()
rt -= t0
span = rt[:, 5].max()
lines = []
P = lines.append
flops = sum(2.0 * n * c * k for n, c, k in shapes)
P("# Distance + argmin kernel: per-workgroup timeline (round 3)\n")
P(f"`python tools/vq_timeline.py` on the debug build (`make -C vq_seg_amd/csrc timeline`), one grouped launch of the bench's three levels "
  f"(N = 131072 / 32768 / 8192 bf16 rows x C = 512 / 1024 / 2048, K = 512): {len(t)} workgroups that did work (of {n_wg} launched), "
  f"{len(np.unique(cu))} distinct CUs seen.\n")
P(f"* launch time by HIP events: {launch_us:.1f} us (= {flops / launch_us / 1e6:.1f} TF/s = {flops / launch_us / 1e6 / 157.3:.3f} of the 157.3 TF/s fp32-MFMA peak; the stamps cost a little)")
P(f"* first workgroup entry -> last workgroup exit: {span:.1f} us  (launch time - span = {launch_us - span:.1f} us: dispatch start-up + end-of-kernel + event overhead)")
dck, drt = ck[:, 5] - ck[:, 0], (rt[:, 5] - rt[:, 0])
mhz = dck / np.maximum(drt, 1e-9)
P(f"* shader clock inside the kernel (s_memtime ticks / s_memrealtime us, per workgroup): median {np.median(mhz):.0f} MHz, p5 {np.percentile(mhz, 5):.0f}, p95 {np.percentile(mhz, 95):.0f}"
  f"  (nominal 2400 MHz; the fp32-MFMA peak is quoted at 2400)")
P("")
P("| level | workgroups | whole (us) | prologue: entry -> first MFMA | main loop | minima (VALU epilogue) | key merge | atomic + exit | ideal loop at 2.4 GHz, 2 waves / SIMD |")
P("|---|---|---|---|---|---|---|---|---|")
names = {0: "C = 2048 (L4)", 1: "C = 1024 (L3)", 2: "C = 512 (L2)"}
for lv in sorted(np.unique(lvl)):
    m = lvl == lv
    d = np.diff(rt[m], axis=1)
    c = {0: 2048, 1: 1024, 2: 512}[int(lv)]
    ideal = (c / 2) * 8 * 64 * 2 / 2400.0                  # MFMAs per wave x 64 cycles x 2 waves sharing the SIMD
    med = lambda v: f"{np.median(v):.2f} (p95 {np.percentile(v, 95):.2f})"
    P(f"| {names[int(lv)]} | {m.sum()} | {med(rt[m, 5] - rt[m, 0])} | {med(d[:, 0])} | {med(d[:, 1])} | {med(d[:, 2])} | {med(d[:, 3])} | {med(d[:, 4])} | {ideal:.1f} |")
P("")
# per CU: the workgroups in start order; occupancy over time
order = np.argsort(rt[:, 0])
gaps, per_cu_busy = [], []
for c_ in np.unique(cu):
    idx = order[cu[order] == c_]
    ends = []                                              # two slots per CU: a new workgroup takes the slot that freed first
    for i in idx:
        if len(ends) < 2:
            ends.append(rt[i, 5])
            continue
        j = int(np.argmin(ends))
        gaps.append(rt[i, 0] - ends[j])
        ends[j] = rt[i, 5]
    loop = (rt[idx, 2] - rt[idx, 1]).sum()
    per_cu_busy.append(loop / (2 * span))
gaps = np.array(gaps)
P(f"* gap on a CU slot between one workgroup's exit and the next one's entry (dispatch): median {np.median(gaps):.2f} us, p95 {np.percentile(gaps, 95):.2f}, "
  f"max {gaps.max():.2f} ({len(gaps)} hand-overs)")
P(f"* fraction of (2 slots x span) that the CUs' workgroups spend inside their main loops: median over CUs {np.median(per_cu_busy):.3f}, min {np.min(per_cu_busy):.3f}")
first_start = np.array([rt[cu == c_, 0].min() for c_ in np.unique(cu)])
last_end = np.array([rt[cu == c_, 5].max() for c_ in np.unique(cu)])
P(f"* start-up ramp: the CUs receive their first workgroup between 0 and {first_start.max():.2f} us (median {np.median(first_start):.2f})")
P(f"* tail: the CUs finish their last workgroup between {last_end.min():.1f} and {last_end.max():.1f} us (median {np.median(last_end):.1f}) -> "
  f"mean idle tail per CU {np.mean(span - last_end):.1f} us = {np.mean(span - last_end) / span * 100:.1f} % of the span")
wg_per_cu = np.array([(cu == c_).sum() for c_ in np.unique(cu)])
P(f"* workgroups per CU: min {wg_per_cu.min()}, median {np.median(wg_per_cu):.0f}, max {wg_per_cu.max()}")
# do the two resident workgroups of a CU run in lockstep?  overlap of their non-loop phases
both_out = 0.0
for c_ in np.unique(cu):
    idx = np.where(cu == c_)[0]
    ev = []
    for i in idx:                                          # intervals in which workgroup i is resident but NOT in its main loop
        ev += [(rt[i, 0], rt[i, 1]), (rt[i, 2], rt[i, 5])]
    ev.sort()
    for a in range(len(ev)):
        for b in range(a + 1, len(ev)):
            if ev[b][0] >= ev[a][1]:
                break
            both_out += min(ev[a][1], ev[b][1]) - ev[b][0]
P(f"* time in which BOTH resident workgroups of a CU are outside their main loops at once (their prologue / epilogue phases coincide: "
  f"nothing feeds that CU's MFMA pipes): {both_out / len(np.unique(cu)):.1f} us per CU = {both_out / len(np.unique(cu)) / span * 100:.1f} % of the span")
out = "\n".join(lines) + "\n"
print(out)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write(out)
    np.save(os.path.splitext(sys.argv[1])[0] + "_raw.npy", t)
