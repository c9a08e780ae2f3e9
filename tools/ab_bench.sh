#!/bin/bash
# A/B of dispatch options on the full bench inside ONE gpurun call (boxes differ by a few percent; runs alternate).
# usage: bash tools/ab_bench.sh "optA=1,optB=2" ["other options"] ...   ("" = defaults)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for opts in "$@"; do
    VQSEG_OPTS="$opts" timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline > /tmp/ab_bench.log 2>&1
    python3 - "$opts" <<'PY'
import json, sys
l = json.loads(open('/tmp/ab_bench.log').read().strip().splitlines()[-1])
print(f"[{sys.argv[1] or 'defaults':60s}] {l['value']:8.2f} img/s  {l['ms_per_step']:7.2f} ms  roofline {l['roofline']['frac']:.4f}", flush=True)
PY
  done
done
