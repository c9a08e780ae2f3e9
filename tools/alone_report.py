"""Which kernels run ALONE (no kernel of another queue in flight) in the steady-state steps of a rocprofv3 kernel trace of bench.py:
   python tools/alone_report.py <kernel_trace.csv> i0 i1        (window: end of step i0 .. end of step i1, as tools/step_timeline.py)
Per kernel name: total time, time alone, time overlapped; then the idle gaps by the kernel that follows them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
marks = [e[1] for e in ev if "confusion" in e[2]]
i0, i1 = int(sys.argv[2]), int(sys.argv[3])
a0, a1 = marks[i0], marks[i1]
ev = [e for e in ev if e[1] > a0 and e[0] < a1]
steps = i1 - i0
# sweep: boundaries
pts = sorted(set([e[0] for e in ev] + [e[1] for e in ev]))
import bisect
starts = sorted(ev, key=lambda e: e[0])
active = []
alone = collections.Counter(); total = collections.Counter(); calls = collections.Counter()
idle_before = collections.Counter()
j = 0
import heapq
cur = []  # (end, name, q)
last_t = pts[0]
for t in pts:
    dt = t - last_t
    if dt > 0:
        if len(cur) == 1:
            alone[cur[0][1]] += dt
        for c in cur:
            total[c[1]] += dt
    cur = [c for c in cur if c[0] > t]
    was_idle = not cur
    while j < len(starts) and starts[j][0] == t:
        cur.append((starts[j][1], starts[j][2], starts[j][3])); calls[starts[j][2]] += 1
        j += 1
    last_t = t
def short(n):
    n = n.replace("void vqseg::", "").replace("vqseg::", "")
    return n[:86]
print(f"window: {steps} steps, {(a1 - a0) / 1e6 / steps:.1f} ms per step; per step: kernel time {sum(total.values()) / 1e6 / steps:.1f} ms, "
      f"of it alone {sum(alone.values()) / 1e6 / steps:.1f} ms")
print(f"{'kernel':86s} {'calls':>6s} {'ms/step':>8s} {'alone':>8s}")
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{short(n):86s} {calls[n] / steps:6.0f} {total[n] / 1e6 / steps:8.2f} {v / 1e6 / steps:8.2f}")
