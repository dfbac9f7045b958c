#!/usr/bin/env python3
"""Development tool: does an HBM-bound LDE hide under VALU-bound Merkle hashing on a second stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stark_rs_amd as s

e1 = s.Engine(s.P2, s.G2, 0)
e2 = s.Engine(s.P2, s.G2, 0)
W, L, lb = 4, 22, 3
N = 1 << (L + lb)
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.integers(0, s.P2, W << L, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
y = torch.empty(W * N, dtype=torch.int32, device="cuda")
cw = torch.from_numpy(rng.integers(0, s.P2, N, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
nodes = torch.empty((2 * N - 1) * 32, dtype=torch.uint8, device="cuda")


def timed(fn, reps=10):
    fn(); e1.sync(); e2.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    e1.sync(); e2.sync()
    return 1e3 * (time.perf_counter() - t0) / reps


lde = lambda: e1.dev_lde(x.data_ptr(), W, L, lb, y.data_ptr())
tree = lambda: e2.dev_merkle_build(cw.data_ptr(), N, nodes.data_ptr())
def both():
    tree(); lde()
print(f"lde alone {timed(lde):.3f} ms, tree alone {timed(tree):.3f} ms, both streams {timed(both):.3f} ms", flush=True)
