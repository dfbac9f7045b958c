#!/usr/bin/env python3
"""Development tool (GPU box): un-instrumented time per call of one transform or extension in the sustained regime (0.3 s of
untimed calls, then >= 1 s timed), for plan sweeps through SMI_NTT_PLAN_<L>.
    SMI_NTT_PLAN_23=8.6,6.6,9.5 python3 tools/plan_time.py lde:20:3:4 [label]      (also ntt:L:batch[:inv])"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import stark_rs_amd as s  # noqa: E402

f = sys.argv[1].split(":")
label = sys.argv[2] if len(sys.argv) > 2 else ""
kind = f[0]
top = int(f[1]) + (int(f[2]) if kind == "lde" else 0)
p = s.P2 if top > 23 else s.P_REF
e = s.Engine(p, s.G2 if p == s.P2 else s.G_REF, 0)
rng = np.random.default_rng(1)
if kind == "lde":
    L, lb, W = int(f[1]), int(f[2]), int(f[3])
    x = torch.from_numpy(rng.integers(0, p, W << L, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
    y = torch.empty(W << (L + lb), dtype=torch.int32, device="cuda")
    run = lambda: e.dev_lde(x.data_ptr(), W, L, lb, y.data_ptr())
else:
    L, batch, inv = int(f[1]), int(f[2]) if len(f) > 2 else 1, "inv" in f
    x = torch.from_numpy(rng.integers(0, p, batch << L, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
    y = torch.empty_like(x)
    run = lambda: e.dev_ntt(x.data_ptr(), y.data_ptr(), L, batch=batch, inverse=inv, offset=3)
for _ in range(3):
    run()
e.sync()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 0.3:
    for _ in range(20):
        run()
    e.sync()
    n += 20
per = (time.perf_counter() - t0) / n
reps = max(50, int(1.0 / per))
t0 = time.perf_counter()
for _ in range(reps):
    run()
e.sync()
print(f"{sys.argv[1]:14s} {label:28s} {1e3 * (time.perf_counter() - t0) / reps:8.4f} ms per call ({reps} calls)", flush=True)
