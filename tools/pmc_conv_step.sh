#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 counter passes over `bench.py` for EVERY convolution kernel of the step (forward, data
# gradient, weight gradient; one counter group per pass, kernel-trace only) -> gpurun_out/pmc_conv_step/{fetch,write}.csv
# (rows of other kernels are dropped on the box).  Summary: tools/summarize_pmc_conv_step.py -> profiles/r04_conv_step_pmc.{json,md}
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_conv_step
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/pmcc_$name
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcc_$name -o run -- \
      python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/$name.log 2>&1
  f=$(ls /tmp/pmcc_$name/*counter_collection.csv | head -1)
  head -1 $f > $OUT/$name.csv
  grep -E "conv3x3_patch_kernel|conv_igemm|conv_wgrad" $f >> $OUT/$name.csv || true
  echo "pass $name: $(wc -l < $OUT/$name.csv) rows"
done
