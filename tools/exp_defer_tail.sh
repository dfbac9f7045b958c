run() { python bench.py --steps 10 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ms_per_step %.4f'%r['ms_per_step'], 'prove', round(r.get('prove_ms',0),3), r.get('prove_stage_ms'))
for k,v in r['roofline']['kernels'].items(): print('    ',k,'%.1f us'%(v['avg_ms']*1e3))
"; }
echo "== default"; run
echo "== defer tw"; SMI_NTT_DEFER_TW=1 run
echo "== tail 0"; SMI_FRI_TAIL=0 run
echo "== tail 1024"; SMI_FRI_TAIL=1024 run
echo "== tail 2048"; SMI_FRI_TAIL=2048 run
