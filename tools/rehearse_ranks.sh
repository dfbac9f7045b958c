#!/bin/bash
# Development tool (GPU box, one card): the N > 1 control flow of bench.py rehearsed with two gloo ranks sharing the card, and the
# in-library RCCL legs at world size 1 (SMI_BENCH_FORCE_DIST).
set -e
export TMPDIR=/tmp
SMI_BENCH_BACKEND=gloo SMI_BENCH_DEVICE=0 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r03_final_rehearsal_2ranks.json 2> gpurun_out/r03_final_rehearsal_2ranks.err || { tail -30 gpurun_out/r03_final_rehearsal_2ranks.err; exit 1; }
python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print({k:(v if not isinstance(v,dict) else {a:b for a,b in list(v.items())[:8]}) for k,v in r.items() if 'mgpu' in k or 'self' in k or 'check' in k})
" gpurun_out/r03_final_rehearsal_2ranks.json
SMI_BENCH_FORCE_DIST=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_final_forcedist_1rank.json 2> gpurun_out/r03_final_forcedist_1rank.err || { tail -30 gpurun_out/r03_final_forcedist_1rank.err; exit 1; }
python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:(v if not isinstance(v,dict) else {a:b for a,b in list(v.items())[:10]}) for k,v in r.items() if 'mgpu' in k or 'self' in k or 'check' in k})
" gpurun_out/r03_final_forcedist_1rank.json
