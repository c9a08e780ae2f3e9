"""The kernels of the CALLER's stream (the queue that carries the loss block, metrics and joins) in one steady-state step of a rocprofv3
kernel trace of bench.py, in time order:  python tools/main_stream_report.py <kernel_trace.csv> step_index
Shows where both networks' streams wait for a serial section."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
marks = [e[1] for e in ev if "confusion" in e[2]]
i = int(sys.argv[2])
a0, a1 = marks[i - 1], marks[i]
ev = [e for e in ev if e[1] > a0 and e[0] < a1]
busy = collections.Counter()
for a, b, n, q in ev: busy[q] += b - a
qs = [q for q, _ in busy.most_common()]
main = qs[2] if len(qs) > 2 else qs[-1]
print(f"step {i}: {(a1 - a0) / 1e6:.1f} ms; queues by kernel time: " + ", ".join(f"{q}: {busy[q] / 1e6:.1f} ms" for q in qs))
others = sorted((a, b) for a, b, n, q in ev if q != main)
def other_busy(t0, t1):
    return sum(max(0, min(b, t1) - max(a, t0)) for a, b in others if b > t0 and a < t1)
last = None
for a, b, n, q in ev:
    if q != main: continue
    n = n.replace("void vqseg::", "").replace("vqseg::", "").replace("void at::native::", "at::")[:70]
    gap = (a - last) / 1e3 if last else 0.0
    print(f"{(a - a0) / 1e6:8.3f} ms  +{gap:7.1f} us gap  {(b - a) / 1e3:7.1f} us  others busy {100 * other_busy(a, b) / max(b - a, 1):5.1f} %  {n}")
    last = b
