#!/usr/bin/env python3
"""Kernel-level timing helper (development tool): per-kernel HIP-event times of NTT / Merkle /
fold launches for a list of configurations.  Usage on the GPU box:
    python tools/kbench.py ntt:20:1 ntt:22:4:inv lde:22:3:4 merkle:23 fold:25 prove:22:3:4
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import stark_rs_amd as s  # noqa: E402


def rnd(n, p):
    rng = np.random.default_rng(1)
    return torch.from_numpy(rng.integers(0, p, n, dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()


def report(eng, label, reps, wall):
    k = eng.profile_read()
    tot = sum(v["total_ms"] for v in k.values())
    print(f"== {label}: wall {1e3 * wall / reps:.4f} ms/iter, kernels {tot / reps:.4f} ms/iter")
    for name, v in sorted(k.items(), key=lambda kv: -kv[1]["total_ms"]):
        avg = v["total_ms"] / v["launches"]
        print(f"   {name:34s} x{v['launches'] // reps:<3d} avg {avg * 1e3:9.1f} us  {v['alg_bytes'] / v['total_ms'] / 1e6:8.1f} GB/s")


def main():
    engs = {}

    def eng_for(p):
        if p not in engs:
            engs[p] = s.Engine(p, s.G2 if p == s.P2 else s.G_REF, 0)
        return engs[p]

    reps = int(os.environ.get("REPS", "20"))
    for spec in sys.argv[1:]:
        f = spec.split(":")
        kind = f[0]
        p = s.P2 if "p2" in f or (kind in ("ntt", "lde", "hostlde", "prove", "merkle", "fold") and int(f[1]) + (int(f[2]) if kind in ("lde", "hostlde", "prove") else 0) > 23) else s.P_REF
        e = eng_for(p)
        if kind == "ntt":
            L, batch, inv = int(f[1]), int(f[2]) if len(f) > 2 else 1, "inv" in f
            x = rnd(batch << L, p); y = torch.empty_like(x)
            run = lambda: e.dev_ntt(x.data_ptr(), y.data_ptr(), L, batch=batch, inverse=inv, offset=3)
        elif kind == "lde":
            L, lb, W = int(f[1]), int(f[2]), int(f[3])
            x = rnd(W << L, p); y = torch.empty(W << (L + lb), dtype=torch.int32, device="cuda")
            run = lambda: e.dev_lde(x.data_ptr(), W, L, lb, y.data_ptr())
        elif kind == "merkle":
            L = int(f[1])
            x = rnd(1 << L, p); y = torch.empty(((2 << L) - 1) * 8, dtype=torch.int32, device="cuda")
            run = lambda: e.dev_merkle_build(x.data_ptr(), 1 << L, y.data_ptr())
        elif kind == "fold":
            L = int(f[1])
            x = rnd(1 << L, p); y = torch.empty(1 << (L - 1), dtype=torch.int32, device="cuda")
            a = torch.tensor([0x0123456789ABCDEF], dtype=torch.int64, device="cuda")
            w = e.prim_nth_root(1 << L)
            run = lambda: e.dev_fri_fold(x.data_ptr(), 1 << L, a.data_ptr(), 3, w, y.data_ptr())
        elif kind == "hostlde":   # host-buffer entry point: u64 in/out over PCIe (the drop-in call)
            L, lb, W = int(f[1]), int(f[2]), int(f[3])
            hx = np.random.default_rng(1).integers(0, p, (W, 1 << L), dtype=np.int64).astype(np.uint64)
            hy = np.zeros((W, 1 << (L + lb)), dtype=np.uint64) if "warm" in f else None   # caller reuses its output buffer
            run = lambda: e.lde(hx, lb, 1, 3, out=hy)
        elif kind == "prove":
            L, lb, W = int(f[1]), int(f[2]), int(f[3])
            x = rnd(W << L, p)
            run = lambda: e.dev_stark_prove(x.data_ptr(), W, L, lb, 32)
        else:
            raise SystemExit(f"unknown spec {spec}")
        for _ in range(3):
            run()
        e.sync()
        e.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        e.sync()
        wall = time.perf_counter() - t0
        report(e, spec, reps, wall)
        e.profile(False)


if __name__ == "__main__":
    main()
