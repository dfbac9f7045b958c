#!/bin/bash
# Development tool (GPU box): plan shapes of the extension's transforms on the headline step (bench.py --no-extras),
# with the column-sharing / deferred-twiddle pass kernels.   bash tools/plans_sweep2.sh
B="python3 bench.py --no-extras --steps 800 --warmup 200"   # sustained regime (r03): the 20-step burst mostly measures the clock ramp
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-28s'%sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),v['avg_ms']*1e3) for n,v in k.items() if 'pass' in n))
" $1 "$2"; }
$B > gpurun_out/ps.json; show gpurun_out/ps.json default
SMI_LDE_TWO_PASS=1 $B > gpurun_out/ps.json; show gpurun_out/ps.json "two-pass extension"
for pl in "10.4,8.5,7.5" "10.4,7.5,8.5" "10.3,8.5,7.5" "9.5,9.5,7.5" "9.5,8.6,8.5" "9.5,8.5,8.6" "9.4,8.5,8.5" "10.4,8.6,7.6"; do
  SMI_NTT_PLAN_25=$pl $B > gpurun_out/ps.json 2>/dev/null; show gpurun_out/ps.json "25: $pl"
done
$B > gpurun_out/ps.json; show gpurun_out/ps.json default
SMI_LDE_TWO_PASS=1 $B > gpurun_out/ps.json; show gpurun_out/ps.json "two-pass extension"
for pl in "11.3,11.3" "11.3,11.2" "11.2,11.2" "8.5,7.6,7.5" "8.6,7.6,7.6" "8.4,7.5,7.5" "7.5,8.5,7.5" "7.5,7.5,8.5"; do
  SMI_NTT_PLAN_22=$pl $B > gpurun_out/ps.json 2>/dev/null; show gpurun_out/ps.json "22: $pl"
done
$B > gpurun_out/ps.json; show gpurun_out/ps.json default
SMI_LDE_TWO_PASS=1 $B > gpurun_out/ps.json; show gpurun_out/ps.json "two-pass extension"
