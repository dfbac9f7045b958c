#!/bin/bash
# Development tool (GPU box): the current build against a previous one kept as stark_rs_amd/build/libstarkmi_p.so
# (SMI_LIB), three alternating rounds of the headline step.   bash tools/exp_ab_lib.sh [tag]
set -e
T=${1:-p}
B="python3 bench.py --no-extras --steps 1000 --warmup 200"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-8s'%sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f(%.0f)'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),v['avg_ms']*1e3,(v.get('copy_only_ms') or 0)*1e3) for n,v in k.items() if 'ntt' in n))
" $1 $2; }
for i in 1 2 3; do
SMI_LIB=$PWD/stark_rs_amd/build/libstarkmi_$T.so $B > gpurun_out/ab_0.json; show gpurun_out/ab_0.json previous
$B > gpurun_out/ab_1.json; show gpurun_out/ab_1.json current
done
