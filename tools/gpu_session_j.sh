set -e
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_j_gputest.log 2>&1 || (tail -30 gpurun_out/r03_j_gputest.log; exit 1)
tail -2 gpurun_out/r03_j_gputest.log
for i in 1 2 3; do
  SMI_MERKLE_FUSE=0 python3 tools/prove_time.py 22 unfused
  python3 tools/prove_time.py 22 fused
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_j_fuse_ab.log
