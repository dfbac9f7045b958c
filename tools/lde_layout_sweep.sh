#!/bin/bash
# Development tool: the two-pass extension for (pass-B tile geometry, intermediate layout) pairs, real kernels and
# copy-only twins (in parentheses), with the generic path of the same box first.
#   bash tools/lde_layout_sweep.sh "2 3" "3 2" ...      (pairs "geo_rq lay_kq"; default: all valid pairs)
run() {
  python bench.py --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r['roofline']['kernels']
print('   ms_per_step %.4f  '%r['ms_per_step'] + '  '.join('%s %.0f(%.0f)'%(n.replace('ntt_pass_kernel','p').replace('_kernel',''),v['avg_ms']*1e3,(v.get('copy_only_ms') or 0)*1e3) for n,v in k.items() if 'lde' in n or '<9' in n or '<8,5,mid' in n or '<8,5,last' in n))
"
}
echo "== generic"; run
if [ $# -eq 0 ]; then set -- "2 2" "2 3" "2 4" "3 1" "3 2" "3 3" "3 4"; fi
for pair in "$@"; do
  set -- $pair
  echo "== geo_rq=$1 lay_kq=$2"; SMI_LDE_TWO_PASS=1 SMI_LDE_GEO=$1 SMI_LDE_LAYOUT=$2 run
done
