#!/usr/bin/env python3
"""Development tool (GPU box): randomized parity sweep against the oracle for a time budget.
    python tools/stress.py [seconds]
NTT / iNTT at random sizes, offsets and paddings on both primes, Merkle trees (element and row
leaves), folds, FRI proofs (byte-identical to the oracle's), polynomial products and divisions,
smi_fri_verify against the oracle's verdict on tampered proofs, the whole prove with column openings
(single-GPU and through the multi-GPU entry points at world size 1) + smi_stark_verify, and the
two-pass extension, many-column extensions (the column-sharing pass kernels)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stark_rs_amd as s  # noqa: E402
from oracle import oracle as o  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    o.build()
    rng = np.random.default_rng(int(os.environ.get("SEED", "12345")))
    from stark_rs_amd.mgpu import MultiGpu
    engs = {s.P_REF: s.Engine(s.P_REF, s.G_REF, 0), s.P2: s.Engine(s.P2, s.G2, 0)}
    gens = {s.P_REF: s.G_REF, s.P2: s.G2}
    mgs = {p_: MultiGpu(e_, 0, 1, min_block=1 << 8) for p_, e_ in engs.items()}
    t0, n_cases = time.time(), 0
    last_note = t0
    while time.time() - t0 < budget:
        p = int(rng.choice([s.P_REF, s.P2]))
        e, g = engs[p], gens[p]
        kind = rng.integers(0, 11)
        if kind == 0:      # inverse transform
            L = int(rng.integers(0, 22))
            n = 1 << L
            v = rng.integers(0, p, n, dtype=np.int64).astype(np.uint64)
            if rng.integers(0, 4) == 0:
                v[:] = p - 1
            off = int(rng.integers(1, p))
            assert np.array_equal(e.intt(v, off), o.fast_intt(v, o.ff_prim_nth_root_g(n, p, g), off, p)), ("intt", p, L, off)
        elif kind == 1:    # zero-padded coset transform
            L = int(rng.integers(0, 22))
            n = 1 << L
            nin = int(rng.integers(1, n + 1))
            v = rng.integers(0, p, nin, dtype=np.int64).astype(np.uint64)
            off = int(rng.integers(1, p))
            assert np.array_equal(e.coset_ntt(v, L, off), o.fast_coset_ntt(v, n, o.ff_prim_nth_root_g(n, p, g), off, p)), ("ntt", p, L, nin, off)
        elif kind == 2 and p == s.P_REF:   # Merkle tree over a codeword, random openings
            L = int(rng.integers(0, 15))
            n = 1 << L
            v = rng.integers(0, p, n, dtype=np.int64).astype(np.uint64)
            t = e.merkle_from_codeword(v)
            want = o.merkle_new(o.leaf_hashes(v))
            assert t.root() == bytes(want[-1]), ("merkle root", L)
            for _ in range(3):
                i = int(rng.integers(0, n))
                assert [bytes(h) for h in t.open(i)] == [bytes(h) for h in o.merkle_open(want, n, i)], ("merkle open", L, i)
            t.free()
        elif kind == 3 and p == s.P_REF:   # fold with an unreduced alpha
            L = int(rng.integers(1, 17))
            n = 1 << L
            v = rng.integers(0, p, n, dtype=np.int64).astype(np.uint64)
            w, off = o.ff_prim_nth_root(n), int(rng.integers(1, p))
            alpha = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
            cfg = o.fri_cfg(w, off, n, 4, 1)
            assert np.array_equal(e.fri_fold(v, alpha, off, w), o.fri_fold_codeword(cfg, v, alpha, off, w)), ("fold", L)
        elif kind == 4 and p == s.P_REF:   # Fri::prove, byte-identical
            L = int(rng.integers(5, 13))
            n = 1 << L
            lb = int(rng.integers(2, 4))
            tt = int(rng.integers(1, min(16, (n >> 4)) + 1))
            w, off = o.ff_prim_nth_root(n), int(rng.integers(1, p))
            coeffs = rng.integers(0, p, max(1, n >> lb), dtype=np.int64).astype(np.uint64)
            cw = o.fast_coset_ntt(coeffs, n, w, off)
            cfg = o.fri_cfg(w, off, n, 1 << lb, tt)
            if o.fri_num_rounds(cfg) == 0:
                continue
            try:
                want, wtop = o.fri_prove(cfg, cw)
            except Exception:
                continue   # a sampling assert of the reference for this (n, t): not a parity case
            got, top = e.fri_prove(e.fri_cfg(w, off, n, 1 << lb, tt), cw)
            assert bytes(got) == want and list(top) == wtop and o.fri_verify(cfg, want), ("prove", L, lb, tt)
        elif kind == 5 and p == s.P_REF:   # products and divisions
            na, nb = int(rng.integers(1, 600)), int(rng.integers(1, 300))
            a = rng.integers(0, p, na, dtype=np.int64).astype(np.uint64)
            b = rng.integers(0, p, nb, dtype=np.int64).astype(np.uint64)
            assert list(e.poly_mul(a, b)) == o.poly_mul(a, b), ("mul", na, nb)
            if b.any():
                q, r = e.poly_div(a, b)
                wq, wr = o.poly_div(a, b)
                trim = lambda c: list(np.trim_zeros(np.array([int(x) for x in c], dtype=np.uint64), "b"))
                assert [int(x) for x in q] == wq and trim(r) == trim(wr), ("div", na, nb)
        elif kind == 6 and p == s.P_REF:   # row-leaf tree
            L, W = int(rng.integers(0, 11)), int(rng.integers(1, 10))
            n = 1 << L
            cols = rng.integers(0, p, (W, n), dtype=np.int64).astype(np.uint64)
            d_cols = e.dev_alloc(W * n * 4)
            e.dev_upload(cols.reshape(-1), d_cols)
            d_nodes = e.dev_alloc((2 * n - 1) * 32)
            e.dev_merkle_build_rows(d_cols, W, n, n, d_nodes)
            got = e.dev_download(d_nodes + (2 * n - 2) * 32, 8).astype(np.uint32).view(np.uint8).tobytes()
            leaves = np.stack([np.frombuffer(o.hash_from_field_elements([int(cols[c, i]) for c in range(W)]), dtype=np.uint8) for i in range(n)])
            assert got == o.merkle_commit(leaves), ("row tree", L, W)
            e.dev_free(d_cols)
            e.dev_free(d_nodes)
        elif kind == 7 and p == s.P_REF:   # Fri::verify: same verdict, reason and values as the oracle on a tampered proof
            L = int(rng.integers(5, 11))
            n = 1 << L
            tt = int(rng.integers(1, min(8, (n >> 4)) + 1))
            w, off = o.ff_prim_nth_root(n), int(rng.integers(1, p))
            cw = o.fast_coset_ntt(rng.integers(0, p, n >> 2, dtype=np.int64).astype(np.uint64), n, w, off)
            cfg = o.fri_cfg(w, off, n, 4, tt)
            if o.fri_num_rounds(cfg) == 0:
                continue
            proof = bytearray(o.fri_prove(cfg, cw)[0])
            if rng.integers(0, 4):
                proof[int(rng.integers(0, len(proof)))] ^= 1 << int(rng.integers(0, 8))
            proof = bytes(proof[:int(rng.integers(0, len(proof)))] if rng.integers(0, 8) == 0 else proof)
            ecfg = e.fri_cfg(w, off, n, 4, tt)
            try:
                want = o.fri_verify(cfg, proof, want_values=True)
            except Exception:
                try:
                    e.fri_verify(ecfg, proof)
                    raise AssertionError(("verify: the reference panics, the device verifier does not", L, tt))
                except s.StarkMiError:
                    pass
            else:
                ok, pv, why = e.fri_verify(ecfg, proof)
                assert (ok, pv) == want and (ok or why == o.fri_last_reject()), ("verify", L, tt, why, o.fri_last_reject())
        elif kind == 8:                    # whole prove with column openings: single GPU == multi-GPU entry points, verifier accepts
            logn, lb, W, tt = int(rng.integers(4, 12)), int(rng.integers(2, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 5))
            n = 1 << logn
            if (n << lb) <= max(1 << lb, 4 * tt) or logn + lb > (23 if p == s.P_REF else 26):
                continue
            cols = rng.integers(0, p, (W, n), dtype=np.int64).astype(np.uint64)
            d_cols = e.dev_alloc(W * n * 4)
            e.dev_upload(cols.reshape(-1), d_cols)
            one = e.dev_stark_prove(d_cols, W, logn, lb, tt, open_columns=True)
            roots, proof, top = mgs[p].stark_prove(d_cols, W, logn, lb, tt, open_columns=True)
            assert roots == [bytes(r) for r in one["column_roots"]] and proof == one["proof"] and top == one["top_indices"], ("mgpu prove", logn, lb, W, tt)
            ok, why = e.stark_verify(proof, roots, W, logn, lb, tt, open_columns=True)
            assert ok, ("stark_verify", why)
            bad = bytearray(proof)
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            try:
                assert not e.stark_verify(bytes(bad), roots, W, logn, lb, tt, open_columns=True)[0], ("stark_verify accepted a flipped bit",)
            except s.StarkMiError:
                pass                       # a flipped length field can be the reference's panic
            e.dev_free(d_cols)
        elif kind == 9:                    # two-pass extension == generic extension
            logn, lb, W = int(rng.integers(20, 23)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
            if logn + lb > (23 if p == s.P_REF else 25):
                continue
            n, N = 1 << logn, 1 << (logn + lb)
            cols = rng.integers(0, p, (W, n), dtype=np.int64).astype(np.uint64)
            d_cols, d_a, d_b = e.dev_alloc(W * n * 4), e.dev_alloc(W * N * 4), e.dev_alloc(W * N * 4)
            e.dev_upload(cols.reshape(-1), d_cols)
            e.dev_lde(d_cols, W, logn, lb, d_a, 1, g)
            e.lde_two_pass(True)
            e.dev_lde(d_cols, W, logn, lb, d_b, 1, g)
            e.lde_two_pass(False)
            assert np.array_equal(e.dev_download(d_a, W * N), e.dev_download(d_b, W * N)), ("two-pass lde", logn, lb, W)
            for d in (d_cols, d_a, d_b):
                e.dev_free(d)
        elif kind == 10:                   # many-column extension: the column-sharing pass kernels (one or more column groups,
            # deferred first-pass twiddles) at launch sizes that select them; two random columns against the oracle
            logn, lb, W = int(rng.integers(18, 22)), int(rng.integers(1, 4)), int(rng.integers(2, 10))
            if logn + lb > (23 if p == s.P_REF else 24):
                continue
            n, N = 1 << logn, 1 << (logn + lb)
            cols = rng.integers(0, p, (W, n), dtype=np.int64).astype(np.uint64)
            toff, loff = int(rng.integers(1, p)), int(rng.integers(1, p))
            d_cols, d_out = e.dev_alloc(W * n * 4), e.dev_alloc(W * N * 4)
            e.dev_upload(cols.reshape(-1), d_cols)
            e.dev_lde(d_cols, W, logn, lb, d_out, toff, loff)
            got = e.dev_download(d_out, W * N).reshape(W, N)
            w, Wn = o.ff_prim_nth_root_g(n, p, g), o.ff_prim_nth_root_g(N, p, g)
            for c in rng.choice(W, 2, replace=False):
                assert np.array_equal(got[c], o.fast_coset_ntt(o.fast_intt(cols[c], w, toff, p), N, Wn, loff, p)), ("lde columns", logn, lb, W, int(c))
            e.dev_free(d_cols)
            e.dev_free(d_out)
        else:
            continue
        n_cases += 1
        if time.time() - last_note > 30:     # a line every half minute: long runs must not look hung
            last_note = time.time()
            print(f"stress: {n_cases} cases so far ({last_note - t0:.0f} s)", flush=True)
    print(f"stress: {n_cases} random cases in {time.time() - t0:.0f} s, all equal to the oracle", flush=True)


if __name__ == "__main__":
    main()
