#!/bin/bash
# Development tool (GPU box): the round's closing run -- whole GPU suite, the driver's bench command, rocprofv3 stats + PMC.
set -e
export TMPDIR=/tmp
TAG=${1:-r03_final}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driverflags.json 2> gpurun_out/${TAG}_bench.err
python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.4g ms_per_step %.4f frac %.3f prove_ms %.3f prove_frac %.3f'%(r['value'], r['ms_per_step'], r['roofline']['frac'], r['prove_ms'], r['prove_roofline']['frac']))
" gpurun_out/${TAG}_bench_driverflags.json
bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile.log; exit 1; }
tail -5 gpurun_out/${TAG}_profile.log
