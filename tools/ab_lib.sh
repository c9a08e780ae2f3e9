#!/bin/bash
# A/B of two builds of the library on the full bench inside one gpurun call: bash tools/ab_lib.sh <other.so>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for lib in "" "$1"; do
    VQSEG_LIB="$lib" timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline > /tmp/ab_lib.log 2>&1
    python3 - "${lib:-in-tree}" <<'PY'
import json, sys
l = json.loads(open('/tmp/ab_lib.log').read().strip().splitlines()[-1])
print(f"[{sys.argv[1][-40:]:40s}] {l['value']:8.2f} img/s  {l['ms_per_step']:7.2f} ms  roofline {l['roofline']['frac']:.4f}", flush=True)
PY
  done
done
