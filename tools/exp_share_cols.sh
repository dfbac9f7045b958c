#!/bin/bash
# Development tool (GPU box): the column-sharing pass kernel (ntt_pass_cols_kernel) against one workgroup per
# (tile, column), and register-bounded tuning builds (stark_rs_amd/build/libstarkmi_<tag>.so, SMI_LIB).
#   bash tools/exp_share_cols.sh [tags...]
set -e
B="python3 bench.py --no-extras --steps 20 --warmup 3"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print(sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','cols'),v['avg_ms']*1e3) for n,v in k.items() if 'pass' in n))
" $1 $2; }
for i in 1 2 3; do
$B > gpurun_out/sc_on.json;  show gpurun_out/sc_on.json default
SMI_NTT_SHARE_COLS=0 $B > gpurun_out/sc_off.json; show gpurun_out/sc_off.json off
for t in "$@"; do SMI_LIB=$PWD/stark_rs_amd/build/libstarkmi_$t.so $B > gpurun_out/sc_$t.json; show gpurun_out/sc_$t.json lib_$t; done
done
