#!/bin/bash
# Development tool (GPU box): rocprofv3 kernel stats + PMC traffic of the LDE bench and of a prove.
#   bash tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/..., summaries via tools/pmc_summary.py
# Counters are collected in their own runs (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); --pmc is never combined with trace domains other
# than --kernel-trace.
set -e
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-extras --in-loop-only --steps 200 --warmup 50"
BENCH2="python3 bench.py --no-extras --in-loop-only --steps 100 --warmup 50"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o lde -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats done"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o lde -- $BENCH > /dev/null 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o lde -- $BENCH > /dev/null 2> $OUT/write.err
echo "write done"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $OUT/valu -o lde -- $BENCH > /dev/null 2> $OUT/valu.err
echo "valu done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prove -o prove -- python3 tools/kbench.py prove:22:3:4 > $OUT/prove.log 2> $OUT/prove.err
echo "prove done"
python3 tools/pmc_summary.py $OUT $TAG
# the opt-in two-pass extension (csrc/lde_core.h): kernel stats + traffic, under its own tag
export SMI_LDE_TWO_PASS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o lde2 -- $BENCH2 > $OUT/bench_stats2.json 2> $OUT/stats2.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch2 -o lde2 -- $BENCH2 > /dev/null 2> $OUT/fetch2.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write2 -o lde2 -- $BENCH2 > /dev/null 2> $OUT/write2.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $OUT/valu2 -o lde2 -- $BENCH2 > /dev/null 2> $OUT/valu2.err
unset SMI_LDE_TWO_PASS
echo "two-pass done"
python3 tools/pmc_summary.py $OUT $TAG two_pass
