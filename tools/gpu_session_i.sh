set -e
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_i_gputest.log 2>&1 || (tail -30 gpurun_out/r03_i_gputest.log; exit 1)
tail -2 gpurun_out/r03_i_gputest.log
for i in 1 2 3; do
  SMI_MERKLE_FUSE=0 python3 tools/prove_time.py 22 unfused
  python3 tools/prove_time.py 22 fused
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_i_fuse_ab.log
python3 bench.py > gpurun_out/r03_i_bench.json 2> gpurun_out/r03_i_bench.err
bash tools/profile.sh r03_i > gpurun_out/r03_i_profile.log 2>&1 || (tail -30 gpurun_out/r03_i_profile.log; exit 1)
bash tools/profile_prove_valu.sh r03_i > gpurun_out/r03_i_profile_valu.log 2>&1 || (tail -30 gpurun_out/r03_i_profile_valu.log; exit 1)
bash tools/profile_stalls.sh r03_i > gpurun_out/r03_i_stalls.log 2>&1 || tail -5 gpurun_out/r03_i_stalls.log
echo done
