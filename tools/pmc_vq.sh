#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 counter passes over `bench.py` for the distance+argmin kernel.
# One counter group per pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), kernel-trace only; rows of other kernels are
# dropped on the box to keep the merged output small.  Output: gpurun_out/$PMC_OUT (default pmc_r2)/{fetch,write,mfma}.csv
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${PMC_OUT:-pmc_r4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o run -- \
      python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/$name.log 2>&1
  f=$(ls /tmp/pmc_$name/*counter_collection.csv | head -1)
  head -1 $f > $OUT/$name.csv
  grep -E "vq_assign_f32_kernel|vq_filter_bf16_kernel|vq_resolve_kernel|vq_rescore_kernel" $f >> $OUT/$name.csv || true
  echo "pass $name: $(wc -l < $OUT/$name.csv) rows"
done
