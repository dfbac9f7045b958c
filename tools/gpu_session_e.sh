set -e
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_e_gputest.log 2>&1 || (tail -30 gpurun_out/r03_e_gputest.log; exit 1)
tail -2 gpurun_out/r03_e_gputest.log
bash tools/exp_ab_prove.sh q > gpurun_out/r03_e_pinned_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r03_e_pinned_ab.log
tools/ubench_mix > gpurun_out/r03_e_ubench_mix.log 2>&1
cat gpurun_out/r03_e_ubench_mix.log
python3 bench.py > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err
tail -c 300 gpurun_out/r03_e_bench.json
