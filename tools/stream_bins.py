"""Per-queue occupancy of ONE steady-state step in time bins, from a rocprofv3 kernel trace of bench.py:
   python tools/stream_bins.py <kernel_trace.csv> step_index [bin_ms]
For each bin: busy fraction of the two busiest queues and the kernel that took most of the bin in each."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
marks = [e[1] for e in ev if "confusion" in e[2]]
i = int(sys.argv[2]); bin_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 2e6
a0, a1 = marks[i - 1], marks[i]
ev = [e for e in ev if e[1] > a0 and e[0] < a1]
qs = [q for q, _ in collections.Counter(e[3] for e in ev).most_common(2)]
nb = int((a1 - a0) / bin_ns) + 1
busy = {q: [0.0] * nb for q in qs}
top = {q: [collections.Counter() for _ in range(nb)] for q in qs}
def short(n):
    n = n.replace("void vqseg::", "").replace("vqseg::", "").replace("_ZN5vqseg", "")
    return n.split("(")[0][:44]
for a, b, n, q in ev:
    if q not in busy: continue
    a, b = max(a, a0), min(b, a1)
    k = int((a - a0) / bin_ns)
    while a < b:
        e = min(b, a0 + (k + 1) * bin_ns)
        busy[q][k] += e - a; top[q][k][short(n)] += e - a
        a = e; k += 1
print(f"step {i}: {(a1 - a0) / 1e6:.1f} ms, queues {qs}")
for k in range(nb):
    line = f"{k * bin_ns / 1e6:6.1f} ms "
    for q in qs:
        t = top[q][k].most_common(1)
        line += f"| {100 * busy[q][k] / bin_ns:5.1f} % {t[0][0] if t else '':44s} "
    print(line)
