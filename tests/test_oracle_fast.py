"""Proves the oracle's fast radix-2 NTT/fold restatements equal the op-for-op
Lagrange / power-sum / exp+xgcd restatements (the reference has no NTT: SURVEY F1),
so they may stand in as the checker at sizes the O(n^3) path cannot reach.  CPU only."""
import numpy as np
import pytest

P = 998244353


def _dom(o, omega, offset, n, p=P):
    out, x = [], offset % p
    for _ in range(n):
        out.append(x); x = o.ff_mul(x, omega, p)
    return out


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 5, 7, 8])
@pytest.mark.parametrize("offset", [1, 3, 17])
def test_fast_intt_equals_lagrange(oracle, logn, offset):
    o = oracle
    n = 1 << logn
    omega = o.ff_prim_nth_root(n)
    vals = o.splitmix64(100 + logn, n) % np.uint64(P)
    ref = o.poly_interpolate_domain(_dom(o, omega, offset, n), vals)
    got = o.fast_intt(vals, omega, offset)
    assert o.poly_eq(got, ref)
    if len(ref):
        assert list(got) == list(ref)          # n coefficients incl. trailing zeros (H8)


def test_fast_intt_cfg1_2p10(oracle):
    """BASELINE configs[0]: 2^10-point forward+inverse on the CPU reference path."""
    o = oracle
    n = 1 << 10
    omega = o.ff_prim_nth_root(n)
    assert omega == 258648936
    vals = o.splitmix64(1, n) % np.uint64(P)
    dom = _dom(o, omega, 1, n)
    coeffs = o.poly_interpolate_domain(dom, vals)           # O(n^3) Lagrange, ~seconds
    assert list(o.fast_intt(vals, omega, 1)) == list(coeffs)
    back = o.poly_eval_domain(coeffs, dom)                  # O(n^2) power sums
    assert list(back) == list(vals)
    assert list(o.fast_coset_ntt(coeffs, n, omega, 1)) == list(vals)


@pytest.mark.parametrize("logd,logN,offset", [(0, 3, 1), (2, 2, 3), (3, 6, 3), (5, 8, 7), (6, 9, 1), (4, 10, 3)])
def test_fast_coset_ntt_equals_eval_domain(oracle, logd, logN, offset):
    o = oracle
    d, N = 1 << logd, 1 << logN
    Omega = o.ff_prim_nth_root(N)
    coeffs = o.splitmix64(7 + logd, d) % np.uint64(P)
    ref = o.poly_eval_domain(coeffs, _dom(o, Omega, offset, N))
    assert list(o.fast_coset_ntt(coeffs, N, Omega, offset)) == list(ref)


def test_fast_paths_second_prime(oracle):
    """P2 = 7*2^26+1: the oracle's arithmetic is p-generic (ff.rs:138-189)."""
    o = oracle
    p, g = o.P2, o.G2
    assert o.ff_exp(g, (p - 1) // 2, p) == p - 1            # g is a non-residue -> generator check
    n, N = 64, 512
    w, W = o.ff_prim_nth_root_g(n, p, g), o.ff_prim_nth_root_g(N, p, g)
    assert o.ff_exp(w, n // 2, p) == p - 1
    vals = o.splitmix64(9, n) % np.uint64(p)
    c = o.poly_interpolate_domain(_dom(o, w, 1, n, p), vals, p)
    assert list(o.fast_intt(vals, w, 1, p)) == list(c)
    ref = o.poly_eval_domain(c, _dom(o, W, g, N, p), p)
    assert list(o.fast_coset_ntt(c, N, W, g, p)) == list(ref)


@pytest.mark.parametrize("logn", [1, 2, 6, 10])
def test_fast_fold_equals_reference_fold(oracle, logn):
    o = oracle
    n = 1 << logn
    omega = o.ff_prim_nth_root(n)
    cfg = o.fri_cfg(omega, 3, n, 4, 1)
    cw = o.splitmix64(11, n) % np.uint64(P)
    for alpha in (0, 1, 5, P - 1, P, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF):
        assert list(o.fast_fold(cw, alpha, 3, omega)) == list(o.fri_fold_codeword(cfg, cw, alpha, 3, omega))
