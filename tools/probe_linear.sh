#!/bin/bash
# Development tool (GPU box): copy-only twins of the LDE's passes with the tile's loads and/or stores made one
# contiguous run (SMI_PROBE_LINEAR bit 0 / bit 1) -- what a tile-contiguous inter-pass layout could buy.
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-30s'%sys.argv[2], ' '.join('%s %.0f'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),(v.get('copy_only_ms') or 0)*1e3) for n,v in k.items() if 'pass' in n))
" $1 "$2"; }
for m in 0 1 2 3 0; do
  SMI_PROBE_LINEAR=$m python3 bench.py --no-extras --steps 300 --warmup 100 > gpurun_out/pl.json 2>/dev/null
  show gpurun_out/pl.json "copy twins, linear mode $m"
done
