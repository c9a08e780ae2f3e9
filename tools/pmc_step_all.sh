#!/bin/bash
# Runs on the GPU box: rocprofv3 FETCH_SIZE / WRITE_SIZE passes over `bench.py` for EVERY kernel of the step -> gpurun_out/pmc_step_all/{fetch,write}.csv
# (columns trimmed on the box: kernel name, counter, value, timestamps).  Summary: tools/summarize_pmc_step_all.py -> profiles/r04_step_traffic.md
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_step_all
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/pmca_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmca_$name -o run -- \
      python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/$name.log 2>&1
  f=$(ls /tmp/pmca_$name/*counter_collection.csv | head -1)
  python3 - "$f" "$OUT/$name.csv" <<'PY'
import csv, sys
rows = csv.DictReader(open(sys.argv[1]))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
for r in rows:
    w.writerow([r["Kernel_Name"][:160], r["Counter_Name"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"]])
PY
  echo "pass $name: $(wc -l < $OUT/$name.csv) rows"
done
