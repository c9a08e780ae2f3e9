"""GPU busy / idle and per-stream occupancy of the steady-state steps from a rocprofv3 kernel trace of `bench.py --no-cpu-baseline`:
   python tools/step_timeline.py <kernel_trace.csv> t0_frac t1_frac      (window: end of step i0 .. end of step i1)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows)
marks = [e[1] for e in ev if "confusion" in e[2]]            # one per CPS step (the step's mIoU)
i0, i1 = int(sys.argv[2]), int(sys.argv[3])                  # window = end of step i0 .. end of step i1 (indices into the marks)
a0, a1 = marks[i0], marks[i1]
print(f"{len(marks)} steps in the trace; window = steps {i0 + 1}..{i1}: {(a1 - a0) / 1e6 / (i1 - i0):.1f} ms per step")
ev = [e for e in ev if e[1] > a0 and e[0] < a1]
busy = 0; cur_a = cur_b = None
for a, b, *_ in ev:
    if cur_b is None or a > cur_b:
        if cur_b is not None: busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else: cur_b = max(cur_b, b)
busy += cur_b - cur_a
span = a1 - a0
per_q = {}
for a, b, n, q in ev: per_q[q] = per_q.get(q, 0) + (b - a)
# time with >= 2 kernels running
pts = sorted([(a, 1) for a, b, *_ in ev] + [(b, -1) for a, b, *_ in ev])
depth = 0; last = pts[0][0]; t_by_depth = {}
for t, d in pts:
    t_by_depth[depth] = t_by_depth.get(depth, 0) + (t - last); last = t; depth += d
print(f"window {span / 1e6:.1f} ms: busy {100 * busy / span:.1f} %; kernel time per queue (ms): " + ", ".join(f"{q}: {v / 1e6:.1f}" for q, v in sorted(per_q.items(), key=lambda kv: -kv[1])[:6]))
print("time with k kernels in flight (ms): " + ", ".join(f"{k}: {v / 1e6:.1f}" for k, v in sorted(t_by_depth.items())[:6]))
