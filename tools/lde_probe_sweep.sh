#!/bin/bash
# Development tool: real and copy-only timings of the two-pass extension's kernels for access-pattern
# variants (SMI_LDE_DBG knobs of csrc/lde_core.h), with the generic three-pass kernels of the same box
# first for reference.
run() {
  python bench.py --no-extras --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ms_per_step %.4f'%r['ms_per_step'])
for k,v in r['roofline']['kernels'].items():
    print('  ',k,'real %.1f us'%(v['avg_ms']*1e3),'copy-only %.1f us'%((v.get('copy_only_ms') or 0)*1e3))
"
}
echo "== generic"; run
for v in "$@"; do echo "== SMI_LDE_DBG=$v"; SMI_LDE_TWO_PASS=1 SMI_LDE_DBG=$v run; done
