#!/usr/bin/env python3
"""Development tool (GPU box): un-instrumented prove of the 2^22 x 4 trace (blowup 8, t = 32) -- median wall time and
stage times (events between the stages only) over REPS proves.  SMI_LIB selects the library build (A/B runs:
tools/exp_ab_prove.sh).   python3 tools/prove_time.py [log_rows] [label]"""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import stark_rs_amd as s  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 22
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("SMI_LIB", "current"))
reps = int(os.environ.get("REPS", "15"))
p = s.P2 if L + 3 > 23 else s.P_REF
e = s.Engine(p, s.G2 if p == s.P2 else s.G_REF, 0)
x = torch.from_numpy(np.random.default_rng(1).integers(0, p, 4 << L, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
for _ in range(3):
    e.dev_stark_prove(x.data_ptr(), 4, L, 3, 32)
walls, stages = [], []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = e.dev_stark_prove(x.data_ptr(), 4, L, 3, 32, timed=True)
    walls.append(1e3 * (time.perf_counter() - t0))
    stages.append(r["stage_ms"])
med = {k: statistics.median(st[k] for st in stages) for k in stages[0]}
print(f"{label:28s} prove {statistics.median(walls):7.3f} ms (min {min(walls):.3f})  " + "  ".join(f"{k} {v:.3f}" for k, v in med.items()), flush=True)
