#!/bin/bash
# Development tool: the two-pass extension for every (pass-B tile geometry, intermediate layout) pair, real kernels,
# with the generic path of the same box first.
run() {
  python bench.py --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r['roofline']['kernels']
print('   ms_per_step %.4f  '%r['ms_per_step'] + '  '.join('%s %.0f(%.0f)'%(n.replace('ntt_pass_kernel','p').replace('_kernel',''),v['avg_ms']*1e3,(v.get('copy_only_ms') or 0)*1e3) for n,v in k.items() if 'lde' in n or '<9' in n or '<8,5,mid' in n or '<8,5,last' in n))
"
}
echo "== generic"; run
for geo in 2 3; do for lay in 1 2 3 4; do
  if [ $((4 - geo)) -le $lay ]; then echo "== geo_rq=$geo lay_kq=$lay"; SMI_LDE_TWO_PASS=1 SMI_LDE_GEO=$geo SMI_LDE_LAYOUT=$lay run; fi
done; done
