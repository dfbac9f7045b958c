#!/bin/bash
# Development tool (GPU box): the first pass's inter-pass twiddles left to the column-sharing middle pass
# (SMI_NTT_DEFER_TW: NTT_TW_SKIP / NTT_TW_IN) against computing them in the first pass; and, deferred, held as 16
# per-thread input multipliers derived once for the columns of a tile (SMI_NTT_TWIN_REGS=1) or re-derived as running
# products per column.   bash tools/exp_defer_cols.sh
set -e
B="python3 bench.py --no-extras --steps 20 --warmup 3"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print(sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','cols'),v['avg_ms']*1e3) for n,v in k.items() if 'pass' in n))
" $1 $2; }
for i in 1 2 3; do
SMI_NTT_DEFER_TW=0 $B > gpurun_out/dc_0.json; show gpurun_out/dc_0.json no_defer
SMI_NTT_DEFER_TW=1 SMI_NTT_TWIN_REGS=1 $B > gpurun_out/dc_1.json; show gpurun_out/dc_1.json defer_held
SMI_NTT_DEFER_TW=1 SMI_NTT_TWIN_REGS=0 $B > gpurun_out/dc_2.json; show gpurun_out/dc_2.json defer_running
done
