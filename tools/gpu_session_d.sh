set -e
export TMPDIR=/tmp
bash tools/exp_ab_prove.sh q > gpurun_out/r03_d_ringadd_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r03_d_ringadd_ab.log
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r03_d_tests.log 2>&1 || (tail -20 gpurun_out/r03_d_tests.log; exit 1)
tail -2 gpurun_out/r03_d_tests.log
bash tools/plans_sweep2.sh > gpurun_out/r03_d_plans_sweep.log 2>&1
cat gpurun_out/r03_d_plans_sweep.log
