#!/usr/bin/env python3
"""Generates tests/golden/golden.json with the CPU oracle (oracle/stark_oracle.c).

The reference ships no fixtures and cannot be built here (Rust, no toolchain), so these vectors
are minted by the op-for-op restatement AFTER it passes the reference's known-answer tests
(tests/test_oracle_kats.py): small inputs/outputs verbatim, SHA-256 of large outputs.  They pin the
oracle against regressions and let the GPU tests check large sizes without re-running the slow
CPU path.  Digest / root / proof values are "parity unpinned" by the reference itself (no value is
asserted anywhere in src/hash.rs, src/merkle.rs, src/fri.rs) -- see DESIGN.md section 5.

    python tests/golden/make_golden.py        # rewrites golden.json (takes ~2 min)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

P, G = o.P_REF, o.G_REF
P2, G2 = o.P2, o.G2


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def vals(seed, n, p=P):
    return o.splitmix64(seed, n) % np.uint64(p)


def main():
    g = {"_about": "minted by oracle/stark_oracle.c via tests/golden/make_golden.py; see that file's docstring"}
    # ---- hash / merkle (small, verbatim)
    g["hash_bytes"] = {m.hex(): o.hash_from_bytes(m).hex() for m in
                       (b"", b"hello", b"hallo", bytes(range(31)), bytes(range(32)), bytes(range(33)), bytes(range(100)))}
    g["hash_field_elements"] = {",".join(map(str, e)): o.hash_from_field_elements(e).hex()
                                for e in ([0], [5], [P - 1], [1, 2, 3, 4, 5], [2 ** 64 - 1])}
    leaves8 = np.stack([np.frombuffer(o.hash_from_bytes(bytes([i])), dtype=np.uint8) for i in range(8)])
    g["merkle_root_8_from_bytes_i"] = o.merkle_commit(leaves8).hex()
    cw = vals(7, 1 << 12)
    nodes = o.merkle_new(o.leaf_hashes(cw))
    g["merkle_codeword_2p12_seed7"] = {"root": bytes(nodes[-1]).hex(), "path_1234": [p.hex() for p in o.merkle_open(nodes, 1 << 12, 1234)]}
    # ---- field constants
    g["roots"] = {str(k): o.ff_prim_nth_root(1 << k) for k in (1, 2, 3, 10, 20, 23)}
    # ---- NTT: small verbatim, large by SHA-256
    v8 = vals(1, 8)
    w8 = o.ff_prim_nth_root(8)
    g["intt_n8_seed1_offset3"] = {"values": [int(x) for x in v8], "coeffs": [int(x) for x in o.fast_intt(v8, w8, 3)]}
    g["ntt"] = {}
    for logn in (10, 16, 20, 23):
        n = 1 << logn
        w = o.ff_prim_nth_root(n)
        v = vals(logn, n)
        g["ntt"][str(logn)] = {"seed": logn, "offset": 3, "intt_sha256": sha(o.fast_intt(v, w, 3)),
                               "coset_ntt_of_first_eighth_sha256": sha(o.fast_coset_ntt(v[: n // 8], n, w, 3))}
    n = 1 << 24
    w = o.ff_prim_nth_root_g(n, P2, G2)
    v = vals(24, n, P2)
    g["ntt_p2_24"] = {"prime": P2, "seed": 24, "offset": 5, "intt_sha256": sha(o.fast_intt(v, w, 5, P2))}
    # ---- BASELINE configs[2]: 2^20 rows x 4 columns, blowup 8, offset g
    lde = {}
    n, N = 1 << 20, 1 << 23
    w, W = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    for c in range(4):
        col = vals(0x5354524B00 + c, n)
        lde[str(c)] = sha(o.fast_coset_ntt(o.fast_intt(col, w, 1), N, W, G))
    g["lde_cfg3_sha256"] = lde
    # ---- FRI: the reference's four cases + a 2^16 case
    fri = {}
    for (nn, e, t, off, coeffs) in ((32, 4, 2, 3, [5]), (64, 4, 3, 7, [5, 3]), (128, 4, 4, 13, [1, 3, 2]),
                                    (256, 8, 5, 17, [1, 2, 5, 3, 7, 4, 1, 2])):
        om = o.ff_prim_nth_root(nn)
        dom = [o.ff_mul(off, o.ff_exp(om, i)) for i in range(nn)]
        cwd = o.poly_eval_domain(coeffs, dom)
        cfg = o.fri_cfg(om, off, nn, e, t)
        proof, top = o.fri_prove(cfg, cwd)
        roots, alphas, last = o.fri_commit_trace(cfg, cwd)
        fri[str(nn)] = {"expansion": e, "t": t, "offset": off, "coeffs": coeffs, "proof_hex": proof.hex(), "top_indices": top,
                        "alphas": alphas, "roots": [bytes(r).hex() for r in roots]}
    nn, e, t, off = 1 << 16, 8, 16, 3
    om = o.ff_prim_nth_root(nn)
    cwd = o.fast_coset_ntt(vals(99, nn // e), nn, om, off)
    cfg = o.fri_cfg(om, off, nn, e, t)
    proof, top = o.fri_prove(cfg, cwd)
    fri[str(nn)] = {"expansion": e, "t": t, "offset": off, "coeff_seed": 99, "proof_sha256": hashlib.sha256(proof).hexdigest(),
                    "proof_len": len(proof), "top_indices": top}
    g["fri"] = fri
    # ---- Polynomial::div (div.rs): small verbatim, a 700 / 325 case by SHA-256 (quotient ‖ trimmed remainder)
    a9, b5 = vals(27, 9), vals(25, 5)
    q, r = o.poly_div(a9, b5)
    g["poly_div_9_by_5"] = {"a": [int(x) for x in a9], "b": [int(x) for x in b5], "q": q, "r": r}
    a, b = vals(3 * 700, 700), vals(5 * 325, 325)
    q, r = o.poly_div(a, b)
    while r and r[-1] == 0:
        r.pop()
    g["poly_div_700_by_325"] = {"seed_a": 2100, "seed_b": 1625, "q_sha256": sha(q), "r_sha256": sha(r), "len_q": len(q), "len_r": len(r)}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    json.dump(g, open(out, "w"), indent=1)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
