B="python3 bench.py --no-extras --steps 1000 --warmup 200"
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=r['roofline']['kernels']
print('%-10s'%sys.argv[2], 'ms_per_step %.4f'%r['ms_per_step'], ' '.join('%s %.0f'%(n.replace('ntt_pass_kernel','').replace('ntt_pass_cols_kernel','c'),v['avg_ms']*1e3) for n,v in k.items() if 'ntt' in n))
" $1 $2; }
for i in 1 2 3; do
  $B > gpurun_out/abn.json 2>/dev/null; show gpurun_out/abn.json default
  for t in mq0 mq0w5 w5; do
    SMI_LIB=$PWD/stark_rs_amd/build/libstarkmi_$t.so $B > gpurun_out/abn.json 2>/dev/null; show gpurun_out/abn.json $t
  done
done
