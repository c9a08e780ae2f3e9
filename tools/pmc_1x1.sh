#!/bin/bash
# usage: pmc_1x1.sh <tag> [VQSEG_OPTS]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_$1
VQSEG_OPTS="$2" timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/p_$1 -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_1x1.py > /dev/null 2>&1
f=$(ls /tmp/p_$1/*counter_collection.csv | head -1)
head -1 $f > $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_1x1_$1.csv
grep -E "conv_igemm_glds|conv3x3_patch" $f >> $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_1x1_$1.csv
