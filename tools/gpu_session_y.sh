#!/bin/bash
# Development tool (GPU box): parity of the row-of-sixteen node hash, the single copy-back and the folds computed by the chunk
# kernel / the fused tail, then A/B against the builds kept as libstarkmi_p.so and libstarkmi_q.so, and a kernel trace.
set -e
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_verify.py tests/test_gpu_mirror.py -x -q > gpurun_out/r03_z_tests.log 2>&1 || { tail -30 gpurun_out/r03_z_tests.log; exit 1; }
tail -2 gpurun_out/r03_z_tests.log
{ echo "== previous = HEAD before this work (libstarkmi_p.so)"; bash tools/exp_ab_prove.sh p;
  echo "== previous = ${QDESC:-libstarkmi_q.so}"; bash tools/exp_ab_prove.sh q;
  echo "== 2^20 x 4 on the reference prime: previous = HEAD before this work"; for i in 1 2 3; do SMI_LIB=$PWD/stark_rs_amd/build/libstarkmi_p.so python3 tools/prove_time.py 20 previous; python3 tools/prove_time.py 20 current; done; } 2>/dev/null > gpurun_out/r03_z_ab.log
cat gpurun_out/r03_z_ab.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_z_prove_trace -o prove -- python3 tools/prove_time.py 22 traced > gpurun_out/r03_z_prove_trace.log 2>&1
python3 tools/trace_gaps.py gpurun_out/r03_z_prove_trace/prove_kernel_trace.csv > gpurun_out/r03_z_prove_timeline.txt
head -32 gpurun_out/r03_z_prove_timeline.txt
