#!/bin/bash
# Runs on the GPU box: per-kernel average durations (rocprofv3 --kernel-trace --stats) of tools/weak_layers.py, one layer per run
#   bash tools/weak_stats.sh [layer ...]   -> gpurun_out/weak_stats/<layer>.txt (kernel, calls, average us)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/weak_stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- l1.conv2 l2.0.conv2 l3.0.conv2 l4.0.conv2 dec4.0 dec4.1
for layer in "$@"; do
  rm -rf /tmp/wst
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wst -o run -- \
      python3 $ROOT/tools/weak_layers.py $layer 5 > $OUT/$layer.log 2>&1
  f=$(ls /tmp/wst/*kernel_stats.csv | head -1)
  python3 - "$f" "$layer" <<'PY' | tee $OUT/$layer.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"== {sys.argv[2]}")
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("conv", "wgrad", "reduce_partials", "reflect", "im2col", "ring_add", "stem7")):
        print(f"  {n[:96]:96s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
done
