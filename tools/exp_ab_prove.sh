#!/bin/bash
# Development tool (GPU box): the current build against a previous one kept as stark_rs_amd/build/libstarkmi_<tag>.so,
# three alternating rounds of the un-instrumented 2^22 x 4 prove.   bash tools/exp_ab_prove.sh [tag]
set -e
T=${1:-p}
for i in 1 2 3; do
  SMI_LIB=$PWD/stark_rs_amd/build/libstarkmi_$T.so python3 tools/prove_time.py 22 previous
  python3 tools/prove_time.py 22 current
done
