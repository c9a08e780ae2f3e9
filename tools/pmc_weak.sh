#!/bin/bash
# Runs on the GPU box: rocprofv3 counter passes (one group per pass, kernel-trace only) over tools/weak_layers.py, one layer per run
# -> gpurun_out/pmc_weak/<layer>.<pass>.csv (conv / wgrad / bn kernels only)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_weak
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for layer in l1.conv2 l2.0.conv2 l3.0.conv2 l4.0.conv2 dec4.0 dec4.1; do
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
    set -- $pass; pname=$1; shift
    rm -rf /tmp/pmcw
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcw -o run -- \
        python3 $ROOT/tools/weak_layers.py $layer 5 > $OUT/$layer.$pname.log 2>&1
    f=$(ls /tmp/pmcw/*counter_collection.csv | head -1)
    head -1 $f > $OUT/$layer.$pname.csv
    grep -E "conv|wgrad|reduce_partials|reflect_fold" $f >> $OUT/$layer.$pname.csv || true
  done
  echo "$layer done: $(tail -1 $OUT/$layer.fetch.log)"
done
