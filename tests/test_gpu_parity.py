"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.
Run on the GPU box with `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P, G = 998244353, 3
P2, G2 = 469762049, 3


@pytest.fixture(scope="module")
def eng():
    import stark_rs_amd as s
    e = s.Engine(P, G, 0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng2():
    import stark_rs_amd as s
    e = s.Engine(P2, G2, 0)
    yield e
    e.close()


def _vals(o, seed, n, p=P):
    return o.splitmix64(seed, n) % np.uint64(p)


def test_scalars(eng, oracle):
    o = oracle
    for lg in (1, 3, 10, 20, 23):
        assert eng.prim_nth_root(1 << lg) == o.ff_prim_nth_root(1 << lg)
    assert eng.inv(123) == o.ff_inv(123) and eng.exp(3, 100) == o.ff_exp(3, 100)
    import stark_rs_amd as s
    with pytest.raises(s.StarkMiError, match="n must be a power of two"):
        eng.prim_nth_root(6)
    with pytest.raises(s.StarkMiError, match="n > 2\\^23 not supported"):
        eng.prim_nth_root(1 << 24)
    with pytest.raises(s.StarkMiError, match="no inverse"):
        eng.inv(0)


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 5, 8])
@pytest.mark.parametrize("offset", [1, 3])
def test_intt_equals_lagrange_oracle(eng, oracle, logn, offset):
    """Polynomial::interpolate_domain, op-for-op oracle (O(n^3)), n <= 256."""
    o = oracle
    n = 1 << logn
    w = o.ff_prim_nth_root(n)
    vals = _vals(o, 40 + logn, n)
    dom = [o.ff_mul(offset, o.ff_exp(w, k)) for k in range(n)]
    ref = o.poly_interpolate_domain(dom, vals)
    got = eng.intt(vals, offset)
    assert o.poly_eq(got, ref)
    assert list(got) == list(ref)


@pytest.mark.parametrize("logd,logN,offset", [(0, 3, 1), (3, 6, 3), (6, 9, 1), (4, 10, 3), (8, 11, 3)])
def test_coset_ntt_equals_eval_domain_oracle(eng, oracle, logd, logN, offset):
    """Polynomial::eval_domain, op-for-op oracle (O(N d))."""
    o = oracle
    d, N = 1 << logd, 1 << logN
    W = o.ff_prim_nth_root(N)
    coeffs = _vals(o, 7 + logd, d)
    dom = [o.ff_mul(offset, o.ff_exp(W, k)) for k in range(N)]
    ref = o.poly_eval_domain(coeffs, dom)
    assert list(eng.coset_ntt(coeffs, logN, offset)) == list(ref)


@pytest.mark.parametrize("logn", [10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23])
def test_ntt_all_plans_vs_fast_oracle(eng, oracle, logn):
    """Every pass plan (small kernel, 2-pass, 3-pass) against the oracle's radix-2 restatement
    (itself proven equal to the O(n^3) path in tests/test_oracle_fast.py)."""
    o = oracle
    n = 1 << logn
    w = o.ff_prim_nth_root(n)
    vals = _vals(o, logn, n)
    assert np.array_equal(eng.intt(vals, 3), o.fast_intt(vals, w, 3))
    nin = n // 8
    assert np.array_equal(eng.coset_ntt(vals[:nin], logn, 3), o.fast_coset_ntt(vals[:nin], n, w, 3))
    assert np.array_equal(eng.coset_ntt(vals, logn, 1), o.fast_coset_ntt(vals, n, w, 1))


def test_ntt_cfg2_2p20_roundtrip_and_sampled_oracle(eng, oracle):
    """BASELINE configs[1]: 2^20-point forward+inverse.  Full-size checks: round trip, and the
    op-for-op oracle's Polynomial::eval at sampled domain points."""
    o = oracle
    n = 1 << 20
    w = o.ff_prim_nth_root(n)
    assert w == 565042129
    vals = _vals(o, 2, n)
    coeffs = eng.intt(vals, 1)
    back = eng.coset_ntt(coeffs, 20, 1)
    assert np.array_equal(back, vals)
    for k in (0, 1, 12345, n // 2, n - 1):
        assert o.poly_eval(coeffs, o.ff_exp(w, k)) == int(vals[k])


def test_second_prime_sizes_above_2p23(eng2, oracle):
    o = oracle
    for logn in (16, 24, 25):
        n = 1 << logn
        w = o.ff_prim_nth_root_g(n, P2, G2)
        vals = _vals(o, logn, n, P2)
        assert np.array_equal(eng2.intt(vals, 31), o.fast_intt(vals, w, 31, P2))
        assert np.array_equal(eng2.coset_ntt(vals[: n // 8], logn, 31), o.fast_coset_ntt(vals[: n // 8], n, w, 31, P2))


@pytest.mark.parametrize("which", ["ref", "p2"])
@pytest.mark.parametrize("logn", [14, 20, 22])
def test_ntt_extreme_values_lazy_ranges(eng, eng2, oracle, which, logn):
    """Constant / alternating inputs of p-1 push every sum path of the lazy butterflies to the
    static bound the kernels track (4p for the reference prime, 8p for p < 2^29): a missed fold
    would wrap mod 2^32 and show up here."""
    o = oracle
    e, p, g = (eng, P, G) if which == "ref" else (eng2, P2, G2)
    n = 1 << logn
    w = o.ff_prim_nth_root_g(n, p, g)
    pats = [np.full(n, p - 1, dtype=np.uint64),
            np.where(np.arange(n) % 2 == 0, p - 1, 0).astype(np.uint64),
            np.where((np.arange(n) >> 4) % 2 == 0, p - 1, 1).astype(np.uint64)]
    for vals in pats:
        assert np.array_equal(e.coset_ntt(vals, logn, 1), o.fast_coset_ntt(vals, n, w, 1, p))
        assert np.array_equal(e.intt(vals, 1), o.fast_intt(vals, w, 1, p))
        assert np.array_equal(e.coset_ntt(vals[: n // 8], logn, 3), o.fast_coset_ntt(vals[: n // 8], n, w, 3, p))


@pytest.mark.parametrize("p,g,logs", [(754974721, 11, (10, 14, 21, 24)), (12289, 11, (3, 12)), (167772161, 3, (13, 20, 25))])
def test_other_primes_run_the_same_kernels(oracle, p, g, logs):
    """The modulus is a run-time parameter (any odd prime < 2^30 with two-adicity >= 12): a prime in
    [2^29, 2^30) takes the 4p lazy range like the reference prime, smaller ones the 8p range, and a
    14-bit prime exercises the bounds of the table sizes.  Parity against the p-generic oracle."""
    import stark_rs_amd as s
    o = oracle
    e = s.Engine(p, g, 0)
    try:
        for logn in logs:
            n = 1 << logn
            w = o.ff_prim_nth_root_g(n, p, g)
            vals = _vals(o, logn, n, p)
            assert np.array_equal(e.intt(vals, 3), o.fast_intt(vals, w, 3, p))
            assert np.array_equal(e.coset_ntt(vals[: max(1, n // 8)], logn, 3), o.fast_coset_ntt(vals[: max(1, n // 8)], n, w, 3, p))
    finally:
        e.close()


@pytest.mark.parametrize("batch", [1, 63, 64, 200])
def test_batched_4096_point_columns(eng, oracle, batch):
    """Columns of 4096 points: fewer than 64 go through the single-workgroup kernel, 64 or more
    through two passes of 64-point lines (planner) -- same results either way."""
    o = oracle
    L = 12
    n = 1 << L
    w = o.ff_prim_nth_root(n)
    cols = _vals(o, 500 + batch, batch * n)
    d_in = eng.dev_alloc(batch * n * 4)
    d_out = eng.dev_alloc(batch * n * 4)
    eng.dev_upload(cols, d_in)
    eng.dev_ntt(d_in, d_out, L, batch=batch, inverse=True, offset=3)
    got = eng.dev_download(d_out, batch * n).reshape(batch, n)
    for c in sorted({0, batch // 3, batch - 1}):
        assert np.array_equal(got[c], o.fast_intt(cols[c * n:(c + 1) * n], w, 3))
    eng.dev_ntt(d_in, d_out, L, batch=batch, offset=5)
    got = eng.dev_download(d_out, batch * n).reshape(batch, n)
    for c in sorted({0, batch - 1}):
        assert np.array_equal(got[c], o.fast_coset_ntt(cols[c * n:(c + 1) * n], n, w, 5))
    eng.dev_free(d_in)
    eng.dev_free(d_out)


@pytest.mark.parametrize("which,L", [("ref", 23), ("p2", 24), ("p2", 25)])
def test_batched_transforms_at_the_recut_sizes(eng, eng2, oracle, which, L):
    """2^23 .. 2^25 points take the digits (8, L - 17, 9) (csrc/ntt_host.h).  A single column goes
    through ntt_pass_kernel at those shapes (checked against the oracle and the golden hashes elsewhere in this file); three
    columns with a coset offset take the column-sharing kernels, ntt_pass_cols_kernel<9,5,last> with its output scale among
    them: every column of the batch must equal the single-column transform of the same input, both directions, and
    the first column the oracle's radix-2 transform."""
    o = oracle
    e, p, g = (eng, P, G) if which == "ref" else (eng2, P2, G2)
    n, batch = 1 << L, 3
    w = o.ff_prim_nth_root_g(n, p, g)
    cols = _vals(o, 900 + L, batch * n, p)
    d_in, d_out, d_one = e.dev_alloc(batch * n * 4), e.dev_alloc(batch * n * 4), e.dev_alloc(n * 4)
    e.dev_upload(cols, d_in)
    for inverse, offset in ((True, 5), (False, 7)):
        e.dev_ntt(d_in, d_out, L, batch=batch, inverse=inverse, offset=offset)
        got = e.dev_download(d_out, batch * n).reshape(batch, n)
        for c in range(batch):
            e.dev_ntt(d_in + c * n * 4, d_one, L, batch=1, inverse=inverse, offset=offset)
            assert np.array_equal(got[c], e.dev_download(d_one, n)), (inverse, c)
        if L == 23:
            want = o.fast_intt(cols[:n], w, offset, p) if inverse else o.fast_coset_ntt(cols[:n], n, w, offset, p)
            assert np.array_equal(got[0], want), inverse
    for ptr in (d_in, d_out, d_one):
        e.dev_free(ptr)


def test_poly_scale(eng, oracle):
    o = oracle
    assert list(eng.poly_scale([1, 2, 3], 2)) == [1, 4, 12]          # mod.rs:439-456
    c = _vals(o, 3, 1000)
    assert list(eng.poly_scale(c, 12345)) == o.poly_scale(c, 12345)


def test_poly_mul_equals_schoolbook_oracle(eng, oracle):
    """Polynomial::mul (mul.rs:6-29) -- the oracle's O(n^2) restatement, incl. the reference KATs."""
    o = oracle
    assert list(eng.poly_mul([1, 1], [1, 1])) == [1, 2, 1]                       # test_mul_linear
    assert list(eng.poly_mul([1, 0, 2], [3, 0, 4])) == [3, 0, 10, 0, 8]          # test_mul_sparse
    assert list(eng.poly_mul([2], [1, 0, 1])) == [2, 0, 2]                       # test_mul_different_degrees
    assert list(eng.poly_mul([P - 1], [2])) == [(2 * (P - 1)) % P]               # test_mul_overflow
    assert len(eng.poly_mul([2, 3], [])) == 0 and len(eng.poly_mul([0, 0], [2, 3])) == 0   # test_mul_zero
    for na, nb in ((5, 9), (100, 157), (700, 325), (1024, 1024)):
        a, b = _vals(o, na, na), _vals(o, nb, nb)
        assert list(eng.poly_mul(a, b)) == o.poly_mul(a, b)


def _trim(c):
    c = [int(v) for v in c]
    while c and c[-1] == 0:
        c.pop()
    return c


def test_poly_div_reference_cases_and_oracle(eng, oracle):
    """Polynomial::div (div.rs:6-42) by power-series inversion on the device.  The reference's own
    tests (div.rs:83-262) with their values, then random sizes against the oracle's restatement of
    the subtraction loop; quotient exact, remainder compared as polynomials (trailing zeros are not
    significant: mod.rs:57-68) and the identity a = q*b + r checked with the device product."""
    import stark_rs_amd as s
    o = oracle
    q, r = eng.poly_div([2, 3, 1], [1, 1])                       # test_division_basic
    assert list(q) == [2, 1] and _trim(r) == []
    q, r = eng.poly_div([1, 0, 1], [1, 1])                       # test_division_with_remainder
    assert len(_trim(q)) == 2 and _trim(r) == [2]
    q, r = eng.poly_div([2, 4, 6], [2])                          # test_division_by_constant
    assert list(q) == [1, 2, 3] and _trim(r) == []
    q, r = eng.poly_div([1, 1], [1, 0, 1])                       # test_division_lower_degree_dividend
    assert len(q) == 0 and list(r) == [1, 1]
    with pytest.raises(s.StarkMiError, match="No division by zero"):   # test_division_by_zero
        eng.poly_div([1, 2], [])
    with pytest.raises(s.StarkMiError, match="No division by zero"):
        eng.poly_div([1, 2], [0, 0])
    q, r = eng.poly_div([], [1, 1])                              # test_division_zero_dividend
    assert len(q) == 0 and _trim(r) == []
    q, r = eng.poly_div([3, 2, 1], [3, 2, 1])                    # test_division_same_polynomials
    assert list(q) == [1] and _trim(r) == []
    q, r = eng.poly_div([7, 14], [7])                            # test_division_with_field_arithmetic
    assert list(q) == [1, 2] and _trim(r) == []
    q, r = eng.poly_div([5, 0, 0, 0, 1, 0, 0], [0, 1, 0])        # leading zeros on both sides
    assert (list(q), _trim(r)) == (o.poly_div([5, 0, 0, 0, 1], [0, 1])[0], _trim(o.poly_div([5, 0, 0, 0, 1], [0, 1])[1]))
    for na, nb in ((9, 5), (157, 100), (700, 325), (1024, 1), (1024, 1024), (1500, 2), (4097, 2049)):
        a, b = _vals(o, 3 * na, na), _vals(o, 5 * nb, nb)
        q, r = eng.poly_div(a, b)
        wq, wr = o.poly_div(a, b) if na * nb <= 300000 else (None, None)
        if wq is not None:
            assert list(q) == wq and _trim(r) == _trim(wr), (na, nb)
        # a = q*b + r
        qb = [int(v) for v in eng.poly_mul(q, b)] if len(q) else []
        back = [(x + (int(r[i]) if i < len(r) else 0)) % P for i, x in enumerate(qb + [0] * (na - len(qb)))]
        assert _trim(back) == _trim(a), (na, nb)
        assert len(_trim(r)) < len(_trim(b))


def test_lde_cfg3_small(eng, oracle):
    """4 columns, blowup 8: interpolate on the subgroup, evaluate on the coset g*<W>."""
    o = oracle
    logn, lb = 12, 3
    n, N = 1 << logn, 1 << (logn + lb)
    w, W = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n) for c in range(4)])
    out = eng.lde(cols, lb, 1, 3)
    for c in range(4):
        coeffs = o.fast_intt(cols[c], w, 1)
        assert np.array_equal(out[c], o.fast_coset_ntt(coeffs, N, W, 3))


def test_non_canonical_input_rejected(eng):
    import stark_rs_amd as s
    v = np.array([1, 2, P, 4], dtype=np.uint64)
    with pytest.raises(s.StarkMiError, match="canonical"):
        eng.intt(v, 1)


@pytest.mark.parametrize("n", [(1 << 20) + 12345, (1 << 24) + 77])
def test_large_host_transfers_pinned_pipeline(eng, n):
    """Transfers of 2^20 elements and more cross PCIe as u32 through two pinned chunks, widened /
    narrowed and range-checked by host threads (api.hip): round trip over one, two and three chunks,
    the reduce flag, and a value >= p in the last chunk."""
    import stark_rs_amd as s
    rng = np.random.default_rng(n)
    v = rng.integers(0, P, n, dtype=np.uint64)
    d = eng.dev_alloc(n * 4)
    eng.dev_upload(v, d)
    assert np.array_equal(eng.dev_download(d, n), v)
    big = v + np.uint64(P) * rng.integers(0, 1 << 30, n, dtype=np.uint64)
    eng.dev_upload(big, d, reduce=True)
    assert np.array_equal(eng.dev_download(d, n), v)
    bad = v.copy()
    bad[n - 3] = P
    with pytest.raises(s.StarkMiError, match="canonical"):
        eng.dev_upload(bad, d)
    eng.dev_upload(v, d)                      # the context stays usable after the rejection
    assert np.array_equal(eng.dev_download(d, n), v)
    eng.dev_free(d)


def test_non_canonical_input_rejected_large(eng):
    import stark_rs_amd as s
    v = np.ones(1 << 20, dtype=np.uint64)
    v[12345] = P + 1
    with pytest.raises(s.StarkMiError, match="canonical"):
        eng.intt(v, 1)


# ------------------------------------------------------------------ hash / merkle
def test_leaf_hashes(eng, oracle):
    o = oracle
    v = np.concatenate([np.array([0, 1, 5, P - 1, 255, 256, 65535, 65536], dtype=np.uint64), _vals(o, 5, 3000)])
    got = eng.hash_leaves(v)
    want = o.leaf_hashes(v)
    assert np.array_equal(got, want)
    assert bytes(got[2]).hex() == "b41399e39a0d1249b4f0318e5e6416ccaa2dffb01519bf1296173be9c10b5f06"


def test_hash_bytes_batch(eng, oracle):
    """smi_hash_bytes_batch: one lane per message, same digests as Hash::from_bytes on each
    (lengths around the 32-byte chunk boundary, the 36-byte seed || counter of index sampling)."""
    o = oracle
    rng = np.random.default_rng(11)
    for ln, n in ((0, 3), (1, 5), (31, 70), (32, 64), (33, 65), (36, 200), (64, 9), (100, 130)):
        msgs = [bytes(rng.integers(0, 256, ln, dtype=np.uint8)) for _ in range(n)]
        got = eng.hash_bytes_batch(msgs)
        assert got == [o.hash_from_bytes(m) for m in msgs], (ln, n)
    assert eng.hash_bytes_batch([]) == []


def test_dev_hash_bytes_transcript_on_device(eng, oracle):
    """smi_dev_hash_bytes: message and digest in device memory (the device-resident Fiat-Shamir
    transcript the multi-GPU loop keeps): digest of every prefix of a run of roots equals
    Hash::from_bytes, and its first 8 bytes are FiatShamir::challenge (src/fiat_shamir.rs:19-25)."""
    import torch
    o = oracle
    rng = np.random.default_rng(21)
    roots = rng.integers(0, 256, 18 * 32, dtype=np.uint8)
    buf = torch.from_numpy(roots).cuda()
    out = torch.empty(32, dtype=torch.uint8, device="cuda")
    for k in range(19):
        eng.dev_hash_bytes(buf.data_ptr(), 32 * k, out.data_ptr())
        eng.sync()
        want = o.hash_from_bytes(bytes(roots[:32 * k]))
        got = bytes(out.cpu().numpy())
        assert got == want, k
        fs = o.FiatShamir()
        fs.absorb(bytes(roots[:32 * k]))
        assert int.from_bytes(got[:8], "little") == fs.challenge()


def test_combine_and_bytes(eng, oracle):
    o = oracle
    rng = np.random.default_rng(3)
    pairs = rng.integers(0, 256, (300, 64), dtype=np.uint8)
    got = eng.hash_combine_pairs(pairs)
    for i in range(300):
        assert bytes(got[i]) == o.hash_combine(bytes(pairs[i, :32]), bytes(pairs[i, 32:]))
    for n in (0, 1, 5, 8, 31, 32, 33, 64, 65, 100, 1000):
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert eng.hash_bytes(m) == o.hash_from_bytes(m)
    assert eng.hash_bytes(b"hello").hex() == "663afaa74185a1693451aa7fd22ac722ff8f89aabc0471f28dc7c2b7354cae8e"


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 18])
def test_merkle_tree_all_levels(eng, oracle, logn):
    """Every level of MerkleTree::new against the oracle.  The sizes cover each way the kernels split a
    tree: thread-local subtrees, chunk workgroups, and inside a chunk the levels hashed one node per
    lane against those hashed over quads of lanes (hash_quad.h: levels of at most 256 nodes)."""
    o = oracle
    n = 1 << logn
    cw = _vals(o, 77 + logn, n)
    tree = eng.merkle_from_codeword(cw)
    nodes = o.merkle_new(o.leaf_hashes(cw))
    off = 0
    for lvl in range(logn + 1):
        cnt = n >> lvl
        assert np.array_equal(tree.level(lvl), nodes[off:off + cnt]), lvl
        off += cnt
    assert tree.root() == bytes(nodes[-1])
    for i in {0, n - 1, n // 2, (n * 3) // 7}:
        path = tree.open(i)
        assert path == o.merkle_open(nodes, n, i)
        assert o.merkle_verify(bytes(nodes[i]), i, path, tree.root())
    tree.free()


def test_merkle_from_digest_leaves_and_panics(eng, oracle):
    import stark_rs_amd as s
    o = oracle
    leaves = np.stack([np.frombuffer(o.hash_from_bytes(bytes([i])), dtype=np.uint8) for i in range(8)])
    assert eng.merkle_commit(leaves).hex() == "d86d7c3c1368c029ff23248875ffb2fb673459897e3dcbd67ac0e09ca4cdd738"
    t = eng.merkle_new(leaves)
    for i in range(8):                                                  # merkle.rs:113-122
        assert o.merkle_verify(bytes(leaves[i]), i, t.open(i), t.root())
    with pytest.raises(s.StarkMiError, match="Index out of bounds"):
        t.open(8)
    with pytest.raises(s.StarkMiError, match="Number of leaves must be power of 2"):
        eng.merkle_new(leaves[:3])
    with pytest.raises(s.StarkMiError, match="Cannot create tree from empty leaves"):
        eng.merkle_new(leaves[:0])


# ------------------------------------------------------------------ fri
@pytest.mark.parametrize("logn", [1, 2, 6, 10, 16])
def test_fold_equals_reference_fold(eng, oracle, logn):
    o = oracle
    n = 1 << logn
    omega = o.ff_prim_nth_root(n)
    cfg = o.fri_cfg(omega, 3, n, 4, 1)
    cw = _vals(o, 11, n)
    for alpha in (0, 1, P - 1, P, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF):
        want = o.fri_fold_codeword(cfg, cw, alpha, 3, omega) if logn <= 10 else o.fast_fold(cw, alpha, 3, omega)
        assert np.array_equal(eng.fri_fold(cw, alpha, 3, omega), want)


def _domain(o, omega, offset, n):
    return [o.ff_mul(offset, o.ff_exp(omega, i)) for i in range(n)]


@pytest.mark.parametrize("n,exp,t,offset,coeffs", [
    (32, 4, 2, 3, [5]),                         # fri.rs:533-563
    (64, 4, 3, 7, [5, 3]),                      # fri.rs:566-601
    (128, 4, 4, 13, [1, 3, 2]),                 # fri.rs:604-646
    (256, 8, 5, 17, [1, 2, 5, 3, 7, 4, 1, 2]),  # fri.rs:649-693
])
def test_fri_prove_reference_cases_byte_identical_and_verified(eng, oracle, n, exp, t, offset, coeffs):
    """The reference's four FRI tests: GPU proof bytes == oracle proof bytes, and the oracle's
    Fri::verify accepts the GPU proof."""
    o = oracle
    omega = o.ff_prim_nth_root(n)
    codeword = o.poly_eval_domain(coeffs, _domain(o, omega, offset, n))
    ocfg = o.fri_cfg(omega, offset, n, exp, t)
    want_proof, want_top = o.fri_prove(ocfg, codeword)
    cfg = eng.fri_cfg(omega, offset, n, exp, t)
    proof, top = eng.fri_prove(cfg, codeword)
    assert top == want_top
    assert proof == want_proof
    assert o.fri_verify(ocfg, proof), o.fri_last_reject()


def test_fri_commit_trace_and_larger_prove(eng, oracle):
    o = oracle
    n, exp, t, offset = 1 << 14, 8, 16, 3
    omega = o.ff_prim_nth_root(n)
    coeffs = _vals(o, 99, n // exp)
    codeword = o.fast_coset_ntt(coeffs, n, omega, offset)
    ocfg = o.fri_cfg(omega, offset, n, exp, t)
    cfg = eng.fri_cfg(omega, offset, n, exp, t)
    roots, alphas, last = eng.fri_commit(cfg, codeword)
    wroots, walphas, wlast = o.fri_commit_trace(ocfg, codeword)
    assert np.array_equal(roots, wroots) and alphas == walphas and np.array_equal(last, wlast)
    proof, top = eng.fri_prove(cfg, codeword)
    wproof, wtop = o.fri_prove(ocfg, codeword)
    assert top == wtop and proof == wproof
    assert o.fri_verify(ocfg, proof), o.fri_last_reject()


def test_fri_commit_run_codewords_and_openings(eng, oracle):
    """Fri::commit's returned codewords (fri.rs:153-155) and openings of the retained per-round trees."""
    o = oracle
    n, exp, t, offset = 1 << 10, 4, 4, 7
    omega = o.ff_prim_nth_root(n)
    codeword = o.fast_coset_ntt(_vals(o, 3, n // exp), n, omega, offset)
    cfg, ocfg = eng.fri_cfg(omega, offset, n, exp, t), o.fri_cfg(omega, offset, n, exp, t)
    roots, alphas, run = eng.fri_commit_run(cfg, codeword)
    wroots, walphas, wlast = o.fri_commit_trace(ocfg, codeword)
    assert np.array_equal(roots, wroots) and alphas == walphas and len(run) == len(wroots)
    cw, w, off = codeword, omega, offset
    for r in range(len(run)):
        assert np.array_equal(run.codeword(r), cw)
        nodes = o.merkle_new(o.leaf_hashes(cw))
        for i in (0, len(cw) // 3, len(cw) - 1):
            assert run.open(r, i) == o.merkle_open(nodes, len(cw), i)
        if r + 1 < len(run):
            cw = o.fri_fold_codeword(ocfg, cw, alphas[r], off, w)
            w, off = o.ff_mul(w, w), o.ff_mul(off, off)
    assert np.array_equal(run.codeword(len(run) - 1), wlast)
    run.free()


def test_fri_panics(eng, oracle):
    import stark_rs_amd as s
    o = oracle
    with pytest.raises(s.StarkMiError, match="Domain length must be power of 2"):
        eng.fri_cfg(3, 3, 48, 4, 2)
    with pytest.raises(s.StarkMiError, match="Expansion factor must be power of 2"):
        eng.fri_cfg(3, 3, 64, 6, 2)
    with pytest.raises(s.StarkMiError, match="Expansion factor must be at least 4"):
        eng.fri_cfg(3, 3, 64, 2, 2)
    cfg = eng.fri_cfg(o.ff_prim_nth_root(32), 3, 32, 4, 2)
    with pytest.raises(s.StarkMiError, match="initial codeword length does not match domain length"):
        eng.fri_prove(cfg, [5] * 16)


# ------------------------------------------------------------------ edge cases
def test_edge_empty_and_tiny_inputs(eng, oracle):
    import stark_rs_amd as s
    o = oracle
    # eval_domain of the empty polynomial is all zeros (eval.rs:36-41 `test_eval_zero_poly`)
    assert not eng.coset_ntt([], 6, 3).any()
    assert not eng.coset_ntt([], 14, 1).any()
    # a single coefficient / a single point
    assert list(eng.coset_ntt([7], 0, 5)) == [7] and list(eng.intt([9], 4)) == [9]
    assert list(eng.coset_ntt([7], 5, 3)) == [7] * 32                      # constant polynomial (eval.rs:44-51)
    # all-zero values interpolate to all-zero coefficients (the mirror maps that to `vec![]`, H8)
    assert not eng.intt(np.zeros(1 << 13, dtype=np.uint64), 3).any()
    # FRI configurations the reference's verify would reject up front: no rounds at all
    with pytest.raises(s.StarkMiError, match="No FRI roots extracted"):
        eng.fri_prove(eng.fri_cfg(o.ff_prim_nth_root(8), 3, 8, 8, 1), [1] * 8)
    # an all-zero codeword is a valid (degree -1) codeword: byte-identical proof
    n = 64
    cfg, ocfg = eng.fri_cfg(o.ff_prim_nth_root(n), 3, n, 4, 3), o.fri_cfg(o.ff_prim_nth_root(n), 3, n, 4, 3)
    assert eng.fri_prove(cfg, [0] * n)[0] == o.fri_prove(ocfg, [0] * n)[0]


@pytest.mark.parametrize("log_blowup", [1, 2, 3, 4])
@pytest.mark.parametrize("logn", [10, 14])
def test_lde_every_blowup(eng, oracle, logn, log_blowup):
    """Zero-padded first passes specialise on the padding factor (2, 4, 8, 16): all of them, on a
    single-kernel size and on a multi-pass size."""
    o = oracle
    n, N = 1 << logn, 1 << (logn + log_blowup)
    w, W = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    cols = np.stack([_vals(o, 900 + c, n) for c in range(2)])
    out = eng.lde(cols, log_blowup, 1, 3)
    for c in range(2):
        assert np.array_equal(out[c], o.fast_coset_ntt(o.fast_intt(cols[c], w, 1), N, W, 3))


def test_ragged_coefficient_counts(eng, oracle):
    """eval_domain with coefficient counts that are not powers of two (zero padding at any length)."""
    o = oracle
    for logN, nc in ((13, 1), (13, 5), (13, 1000), (13, 8191), (16, 8193), (16, 65535)):
        N = 1 << logN
        W = o.ff_prim_nth_root(N)
        c = _vals(o, nc, nc)
        assert np.array_equal(eng.coset_ntt(c, logN, 3), o.fast_coset_ntt(c, N, W, 3)), (logN, nc)


def test_max_size_second_prime_2p26_sampled(eng2, oracle):
    """The largest domain of the second prime (BASELINE configs[3] size): 2^26 points, checked by
    the round trip and by the op-for-op oracle's Polynomial::eval at sampled points."""
    o = oracle
    logn = 26
    n = 1 << logn
    w = o.ff_prim_nth_root_g(n, P2, G2)
    coeffs = _vals(o, 26, 1 << 16, P2)                      # degree < 2^16 keeps the oracle's eval cheap
    ev = eng2.coset_ntt(coeffs, logn, 5)
    for k in (0, 1, 3, n // 3, n - 1):
        assert o.poly_eval(coeffs, o.ff_mul(5, o.ff_exp(w, k, P2), P2), P2) == int(ev[k])
    back = eng2.intt(ev, 5)
    assert np.array_equal(back[: 1 << 16], coeffs) and not back[1 << 16:].any()


def test_fri_prove_while_the_scale_table_cache_wraps(eng, oracle):
    """The context caches at most 96 x^-1 / scale tables and drops the oldest half when full.  The fused FRI
    tail (csrc/hash.hip, fri_tail_kernel) holds one table per remaining fold until its single launch, so an
    eviction while it collects them would leave it reading freed memory (found by tools/stress.py: an
    intermittent wrong proof).  40 proves with fresh offsets wrap the cache several times at every alignment."""
    o = oracle
    n, exp, t = 1 << 12, 4, 4
    omega = o.ff_prim_nth_root(n)
    coeffs = o.splitmix64(3, n // exp) % np.uint64(P)
    for k in range(40):
        offset = 3 + 2 * k
        cw = o.fast_coset_ntt(coeffs, n, omega, offset)
        want, wtop = o.fri_prove(o.fri_cfg(omega, offset, n, exp, t), cw)
        got, top = eng.fri_prove(eng.fri_cfg(omega, offset, n, exp, t), cw)
        assert bytes(got) == want and list(top) == wtop, k
