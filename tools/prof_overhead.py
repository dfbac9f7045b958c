#!/usr/bin/env python3
"""Development tool: wall time per LDE step with and without the per-kernel HIP-event brackets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stark_rs_amd as s

e = s.Engine(s.P2, s.G2, 0)
W, L, lb = 4, 22, 3
x = torch.from_numpy(np.random.default_rng(1).integers(0, s.P2, W << L, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
y = torch.empty(W << (L + lb), dtype=torch.int32, device="cuda")
for prof in (False, True, False, True):
    e.profile(prof)
    for _ in range(5):
        e.dev_lde(x.data_ptr(), W, L, lb, y.data_ptr())
    e.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        e.dev_lde(x.data_ptr(), W, L, lb, y.data_ptr())
    e.sync()
    dt = (time.perf_counter() - t0) / 50
    if prof:
        e.profile_read()
    print(f"profile={prof}: {dt * 1e3:.4f} ms/step", flush=True)
