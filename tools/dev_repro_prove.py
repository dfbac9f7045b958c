#!/usr/bin/env python3
"""Development tool (GPU box): smi_dev_stark_prove vs smi_mgpu_stark_prove (world 1) vs the oracle's composition for
small shapes; prints which of (roots, FRI bytes, openings, indices) differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import stark_rs_amd as s  # noqa: E402
from stark_rs_amd.mgpu import MultiGpu  # noqa: E402
from oracle import oracle as o  # noqa: E402
from conftest import column_openings_bytes  # noqa: E402

o.build()
P, G = s.P_REF, s.G_REF
e = s.Engine(P, G, 0)
mg = MultiGpu(e, 0, 1, min_block=1 << 8)
bad = 0
for (logn, lb, W, t) in [(8, 2, 1, 2), (8, 2, 2, 2), (8, 3, 1, 2), (6, 2, 1, 1), (9, 2, 1, 3), (8, 2, 1, 1), (10, 2, 3, 4), (7, 3, 5, 2)]:
    for seed in range(6):
        n, N = 1 << logn, 1 << (logn + lb)
        rng = np.random.default_rng(seed)
        cols = rng.integers(0, P, (W, n), dtype=np.int64).astype(np.uint64)
        d = e.dev_alloc(W * n * 4)
        e.dev_upload(cols.reshape(-1), d)
        one = e.dev_stark_prove(d, W, logn, lb, t, open_columns=True)
        roots, proof, top = mg.stark_prove(d, W, logn, lb, t, open_columns=True)
        w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
        lde = [o.fast_coset_ntt(o.fast_intt(cols[c], w, 1), N, Wn, G) for c in range(W)]
        fs, weights, want_roots = o.FiatShamir(), [], []
        for c in range(W):
            want_roots.append(o.merkle_commit(o.leaf_hashes(lde[c])))
            fs.absorb(want_roots[-1])
            weights.append(fs.challenge() % P)
        cw = np.zeros(N, dtype=object)
        for c in range(W):
            cw = (cw + lde[c].astype(object) * weights[c]) % P
        ocfg = o.fri_cfg(Wn, G, N, 1 << lb, t)
        wfri, wtop = o.fri_prove(ocfg, cw.astype(np.uint64))
        wopen = column_openings_bytes(o, lde, wtop, N)
        def cmp(name, pr, tp, rt):
            global bad
            f = pr[:len(wfri)] == wfri
            op = pr[len(wfri):] == wopen
            if not (f and op and tp == wtop and rt == [bytes(r) for r in want_roots]):
                bad += 1
                firstdiff = next((i for i in range(min(len(pr), len(wfri) + len(wopen))) if pr[i] != (wfri + wopen)[i]), None)
                print(f"MISMATCH {name} shape={(logn, lb, W, t)} seed={seed}: fri={f} openings={op} top={tp == wtop} roots={rt == [bytes(r) for r in want_roots]} "
                      f"len={len(pr)} want={len(wfri) + len(wopen)} firstdiff={firstdiff} (fri len {len(wfri)})")
        cmp("single", one["proof"], one["top_indices"], [bytes(r) for r in one["column_roots"]])
        cmp("mgpu", proof, top, roots)
        e.dev_free(d)
print("done, mismatches:", bad)
