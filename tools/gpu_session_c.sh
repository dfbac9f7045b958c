set -e
export TMPDIR=/tmp
tools/ubench_mix > gpurun_out/r03_c_ubench_mix.log 2>&1
cat gpurun_out/r03_c_ubench_mix.log
bash tools/profile.sh r03_c > gpurun_out/r03_c_profile.log 2>&1 || (tail -30 gpurun_out/r03_c_profile.log; exit 1)
bash tools/profile_prove_valu.sh r03_c > gpurun_out/r03_c_profile_valu.log 2>&1 || (tail -30 gpurun_out/r03_c_profile_valu.log; exit 1)
echo "profiles done"
SEED=3101 python3 tools/stress.py 240 > gpurun_out/r03_c_stress.log 2>&1 || (tail -30 gpurun_out/r03_c_stress.log; exit 1)
tail -3 gpurun_out/r03_c_stress.log
