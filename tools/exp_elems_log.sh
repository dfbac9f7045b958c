#!/bin/bash
# Development tool (GPU box): from how many leaves on a tree built from codeword elements should start with the
# four-leaves-per-lane kernel instead of the chunk kernel (SMI_MERKLE_ELEMS_LOG; 19 = one chunk workgroup per CU, the r02 rule).
set -e
export TMPDIR=/tmp
SMI_MERKLE_ELEMS_LOG=16 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_verify.py -x -q > gpurun_out/r03_elems_tests.log 2>&1 || { tail -30 gpurun_out/r03_elems_tests.log; exit 1; }
tail -2 gpurun_out/r03_elems_tests.log
{ for i in 1 2 3; do for l in 19 18 17 16 15; do SMI_MERKLE_ELEMS_LOG=$l python3 tools/prove_time.py 22 "elems_log $l"; done; done
  for i in 1 2 3; do for l in 19 18 17 16 15; do SMI_MERKLE_ELEMS_LOG=$l python3 tools/prove_time.py 20 "elems_log $l (2^20 x 4)"; done; done; } 2>/dev/null > gpurun_out/r03_elems_log_ab.log
cat gpurun_out/r03_elems_log_ab.log
