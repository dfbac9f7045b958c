set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
echo "--- default"; python tools/kbench.py ntt:25:4 ntt:22:4:inv ntt:20:1 ntt:23:1 2>&1 | grep -v amdgpu.ids
for pl in "9.5,8.5,8.4" "9.4,8.5,8.5" "9.5,8.6,8.5" "8.6,8.6,9.4" "8.5,8.5,9.4" "8.6,8.5,9.3" "7.6,6.6,6.6,6.6"; do echo "--- 25: $pl"; SMI_NTT_PLAN_25=$pl python tools/kbench.py ntt:25:4 2>&1 | grep -v amdgpu.ids; done
for pl in "10.2,10.2" "10.4,10.3" "10.4,10.4" "10.3,10.3" "7.5,7.5,6.6" "7.6,7.5,6.6" "8.6,6.6,6.6"; do echo "--- 20: $pl"; SMI_NTT_PLAN_20=$pl python tools/kbench.py ntt:20:1 2>&1 | grep -v amdgpu.ids; done
for pl in "8.5,7.6,7.5" "8.6,7.6,7.5" "8.4,7.5,7.5" "8.6,7.6,7.6"; do echo "--- 22: $pl"; SMI_NTT_PLAN_22=$pl python tools/kbench.py ntt:22:4:inv 2>&1 | grep -v amdgpu.ids; done
