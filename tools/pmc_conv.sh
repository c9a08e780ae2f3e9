#!/bin/bash
# Runs on the GPU box: PMC passes over tools/ab_conv.py (3x3 decoder layers, patch kernel) -> gpurun_out/pmc_conv/*.csv (conv rows only)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcc_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmcc_$i -o run -- \
      python3 $ROOT/tools/ab_conv.py "" > $OUT/pass$i.log 2>&1
  f=$(ls /tmp/pmcc_$i/*counter_collection.csv | head -1)
  head -1 $f > $OUT/pass$i.csv
  grep "conv3x3_patch_kernel" $f >> $OUT/pass$i.csv || true
  echo "pass $i: $(wc -l < $OUT/pass$i.csv) rows"
done
