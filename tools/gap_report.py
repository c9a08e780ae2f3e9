"""Largest idle gaps of the GPU (no kernel of any stream running) in the last `frac` of a rocprofv3 kernel trace, with the
kernels that end before / start after each gap:  python tools/gap_report.py <kernel_trace.csv> [frac] [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows)
t_end = max(e[1] for e in ev)
t0 = ev[0][0] + (t_end - ev[0][0]) * (1 - frac)
ev = [e for e in ev if e[1] > t0]
gaps, cur_end, last = [], None, None
for a, b, name in ev:
    if cur_end is not None and a > cur_end:
        gaps.append((a - cur_end, last, name, cur_end))
    if cur_end is None or b > cur_end:
        cur_end, last = b, name
tot = sum(g[0] for g in gaps)
print(f"span {(t_end - t0) / 1e6:.1f} ms, idle {tot / 1e6:.1f} ms in {len(gaps)} gaps")
hist = {}
for g, before, after, _ in gaps:
    k = (before.split('(')[0][-40:], after.split('(')[0][-40:])
    h = hist.setdefault(k, [0, 0])
    h[0] += g
    h[1] += 1
for (b, a), (g, n) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{g / 1e3:9.1f} us in {n:5d} gaps  after [{b}]  before [{a}]")
