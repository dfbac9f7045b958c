#!/bin/bash
# Development tool: copy-only timings of pass A / pass B access-pattern variants (SMI_LDE_DBG knobs of
# csrc/lde_core.h).  Prints kernel -> copy-only ms for each variant.
for v in 0 256 512 768 1024 1280 65536 196608 262144 327680; do
  echo "== SMI_LDE_DBG=$v"
  SMI_LDE_DBG=$v python bench.py --no-extras --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in r['roofline']['kernels'].items():
    if k.startswith('lde_'): print('  ',k,'real %.1f us'%(v['avg_ms']*1e3),'copy-only %.1f us'%((v.get('copy_only_ms') or 0)*1e3))
"
done
