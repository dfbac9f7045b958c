set -e
export TMPDIR=/tmp
bash tools/exp_ab_prove.sh p > gpurun_out/r03_b_hash_ab.log 2>&1
cat gpurun_out/r03_b_hash_ab.log
python3 -m pytest tests/test_gpu_parity.py -x -q -k "hash or merkle or tree or leaf" > gpurun_out/r03_b_hashtests.log 2>&1 || (tail -20 gpurun_out/r03_b_hashtests.log; exit 1)
tail -2 gpurun_out/r03_b_hashtests.log
REPS=5 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_b_prove_trace -o prove -- python3 tools/prove_time.py 22 traced > gpurun_out/r03_b_prove_trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_b_lde_trace -o lde -- python3 bench.py --no-extras --in-loop-only --steps 600 --warmup 0 > gpurun_out/r03_b_lde_trace.json 2> gpurun_out/r03_b_lde_trace.err
python3 bench.py --no-extras > gpurun_out/r03_b_bench_noextras.json 2> /dev/null
tail -c 400 gpurun_out/r03_b_bench_noextras.json
