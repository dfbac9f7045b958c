#!/bin/bash
# Development tool (GPU box): SQ_INSTS_VALU per wave of the prove's kernels.  bash tools/profile_prove_valu.sh <tag>
set -e
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPS=3 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/prove_valu -o prove -- python3 tools/kbench.py prove:22:3:4 > $OUT/prove_valu.log 2> $OUT/prove_valu.err
python3 tools/pmc_summary.py $OUT $TAG prove
