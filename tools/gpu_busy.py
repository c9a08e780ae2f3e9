"""GPU busy fraction from a rocprofv3 kernel trace: length of the union of kernel intervals / span, over the last
`frac` of the trace (the timed steps).  usage: python tools/gpu_busy.py <kernel_trace.csv> [frac]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t_end = iv[-1][1]
t0 = iv[0][0] + (t_end - iv[0][0]) * (1 - frac)
iv = [(max(a, t0), b) for a, b in iv if b > t0]
busy, cur_a, cur_b = 0, None, None
gaps = []
for a, b in iv:
    if cur_b is None or a > cur_b:
        if cur_b is not None:
            busy += cur_b - cur_a
            gaps.append(a - cur_b)
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
span = iv[-1][1] - iv[0][0]
gaps.sort(reverse=True)
print(f"span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms = {100 * busy / span:.1f} %, {len(iv)} kernels, "
      f"{len(gaps)} idle gaps, total idle {sum(gaps) / 1e6:.1f} ms, largest {[round(g / 1e3) for g in gaps[:8]]} us")
