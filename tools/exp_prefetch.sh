#!/bin/bash
# Development tool (GPU box): the middle pass with the next column's loads issued under the current column's stores
# (tuning builds -DSMI_COLS_PREFETCH=1 kept as stark_rs_amd/build/libstarkmi_pf.so / _pfmq0.so) against the default.
B="python3 bench.py --no-extras --steps 1000 --warmup 200"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-10s'%sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),v['avg_ms']*1e3) for n,v in k.items() if 'ntt' in n))
" $1 $2; }
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -k "ntt or lde or plan or transform or cfg" 2>&1 | tail -2
for i in 1 2 3; do
  $B > gpurun_out/abn.json 2>/dev/null; show gpurun_out/abn.json mid+last
  for t in nopf pf; do
    SMI_LIB=$PWD/stark_rs_amd/build/libstarkmi_$t.so $B > gpurun_out/abn.json 2>/dev/null; show gpurun_out/abn.json $t
  done
done
