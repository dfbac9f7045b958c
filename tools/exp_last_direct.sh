#!/bin/bash
# Development tool (GPU box): the last pass's first radix-16 step on its load registers (SMI_NTT_LAST_DIRECT=1,
# NTT_LAST_DIRECT) against the transposition through LDS; kernel times with their copy-only twins.
set -e
B="python3 bench.py --no-extras --steps 20 --warmup 3"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-8s'%sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f(%.0f)'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),v['avg_ms']*1e3,(v.get('copy_only_ms') or 0)*1e3) for n,v in k.items() if 'pass' in n))
" $1 $2; }
for i in 1 2 3; do
SMI_NTT_LAST_DIRECT=0 $B > gpurun_out/ld_0.json; show gpurun_out/ld_0.json lds
SMI_NTT_LAST_DIRECT=1 $B > gpurun_out/ld_1.json; show gpurun_out/ld_1.json direct
done
