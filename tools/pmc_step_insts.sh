#!/bin/bash
# One rocprofv3 counter pass over two bench steps: SQ_INSTS_VALU / SALU / LDS and SQ_WAVES per kernel dispatch -> which kernels are
# bound by instruction issue rather than by memory or MFMA (tools/pmc_step_insts.py summarises).  Output: gpurun_out/$1.csv
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_step
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/p_step -o run -- \
    python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
f=$(ls /tmp/p_step/*counter_collection.csv | head -1)
python3 - "$f" "$GRAFT_REPO_ROOT/gpurun_out/$1.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
        acc[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "total_ns", "waves", "valu", "salu", "lds"])
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
        w.writerow([k, cnt[k], int(c["ns"]), int(c["SQ_WAVES"]), int(c["SQ_INSTS_VALU"]), int(c["SQ_INSTS_SALU"]), int(c["SQ_INSTS_LDS"])])
PY
