#!/bin/bash
# Development tool (GPU box): the headline step with the inter-pass buffers in natural order (SMI_NTT_TILE_MAJOR=0, in place)
# and in the consuming pass's tile-major order (default), three alternating rounds.   bash tools/exp_tile_major.sh
B="python3 bench.py --no-extras --steps 1000 --warmup 200"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-12s'%sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f(%.0f)'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),v['avg_ms']*1e3,(v.get('copy_only_ms') or 0)*1e3) for n,v in k.items() if 'ntt' in n))
" $1 $2; }
for i in 1 2 3; do
  for m in 0 1 2 3; do
    SMI_NTT_TILE_MAJOR=$m $B > gpurun_out/tm_$m.json 2>/dev/null; show gpurun_out/tm_$m.json "mode-$m"
  done
done
