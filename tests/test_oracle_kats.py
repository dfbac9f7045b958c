"""Pins the CPU oracle against every known-answer test the reference holds for the
hot path.  Each test cites the reference test it transliterates.
CPU only."""
import numpy as np
import pytest

P = 998244353


# ------------------------------------------------------------------ ff.rs
def test_ff_add_sub_mul_kats(oracle):
    o = oracle
    assert o.ff_add(100, 200) == 300                      # ff.rs:346-356
    assert o.ff_add(P - 1, 5) == 4                        # ff.rs:359-365
    assert o.ff_add(123, 0) == 123 and o.ff_add(0, 123) == 123
    assert o.ff_sub(200, 100) == 100                      # ff.rs:393-403
    assert o.ff_sub(5, 10) == P - 5                       # ff.rs:406-412
    assert o.ff_sub(0, 123) == P - 123                    # ff.rs:415-425
    assert o.ff_mul(123, 456) == (123 * 456) % P          # ff.rs:428-438
    assert o.ff_mul(1000000, 2000000) == 2000000000000 % P  # ff.rs:467-473
    assert o.ff_mul(123, 0) == 0 and o.ff_mul(123, 1) == 123


def test_ff_neg_inv_div(oracle):
    o = oracle
    assert o.ff_neg(100) == P - 100                       # ff.rs:488-497
    assert o.ff_neg(0) == 0                               # ff.rs:500-505
    assert o.ff_add(123, o.ff_neg(123)) == 0
    for a in (123, 1, P - 1):                             # ff.rs:525-549
        assert o.ff_mul(a, o.ff_inv(a)) == 1
    assert o.ff_inv(1) == 1
    with pytest.raises(o.OraclePanic, match="no inverse"):  # ff.rs:552-557
        o.ff_inv(0)
    c = o.ff_div(1000, 25)                                # ff.rs:560-571
    assert o.ff_mul(c, 25) == 1000
    assert o.ff_div(123, 1) == 123
    with pytest.raises(o.OraclePanic, match="no division by zero"):  # ff.rs:583-589
        o.ff_div(100, 0)
    assert o.ff_inv(2) == 499122177                       # SURVEY 8c constants
    assert o.ff_inv(1 << 20) == 998243401
    assert o.ff_inv(1 << 23) == 998244234


def test_ff_exp_roots(oracle):
    o = oracle
    assert o.ff_exp(3, 2) == 9                            # ff.rs:592-596
    assert o.ff_exp(123, 0) == 1 and o.ff_exp(123, 1) == 123
    assert o.ff_exp(2, 10) == 1024                        # ff.rs:623-628
    assert o.ff_g() == 3                                  # ff.rs:631-635
    with pytest.raises(o.OraclePanic):                    # ff.rs:638-642
        o.ff_g(2147483647)
    for i in range(1, 11):                                # ff.rs:652-660
        n = 1 << i
        assert o.ff_exp(o.ff_prim_nth_root(n), n) == 1
    r8 = o.ff_prim_nth_root(8)                            # ff.rs:663-671 primitivity
    assert all(o.ff_exp(r8, i) != 1 for i in range(1, 8))
    r2, r4 = o.ff_prim_nth_root(2), o.ff_prim_nth_root(4)
    assert len({r2, r4, r8}) == 3
    # values computed from the reference's formula
    assert r2 == P - 1 and r4 == 911660635 and r8 == 372528824
    assert o.ff_prim_nth_root(1 << 20) == 565042129
    assert o.ff_prim_nth_root(1 << 23) == 15311432
    assert o.ff_prim_nth_root(1024) == 258648936
    with pytest.raises(o.OraclePanic):                    # ff.rs:692-696
        o.ff_prim_nth_root(8, 2147483647)
    with pytest.raises(o.OraclePanic, match="n must be a power of two"):  # ff.rs:699-703
        o.ff_prim_nth_root(6)
    with pytest.raises(o.OraclePanic, match="n > 2\\^23 not supported"):  # ff.rs:706-710
        o.ff_prim_nth_root(1 << 24)


def test_ff_sample(oracle):
    o = oracle
    assert o.ff_sample(b"") == 0                          # ff.rs:713-717
    assert o.ff_sample(bytes([42])) == 42                 # ff.rs:720-724
    assert o.ff_sample(bytes([1, 2, 3])) < P
    assert o.ff_sample(bytes([1, 2, 3, 4])) != o.ff_sample(bytes([1, 2, 3, 5]))
    assert o.ff_sample(bytes(range(100))) < P and o.ff_sample(bytes([255] * 10)) < P


def test_ff_properties(oracle):
    o = oracle                                            # ff.rs:766-790
    a, b, c = 123, 456, 789
    assert o.ff_add(o.ff_add(a, b), c) == o.ff_add(a, o.ff_add(b, c))
    assert o.ff_mul(o.ff_mul(a, b), c) == o.ff_mul(a, o.ff_mul(b, c))
    assert o.ff_mul(a, o.ff_add(b, c)) == o.ff_add(o.ff_mul(a, b), o.ff_mul(a, c))


def test_xgcd(oracle):
    g, x, y = oracle.xgcd(240, 46)                        # utils.rs:3-13
    assert g == 2 and 240 * x + 46 * y == 2
    g, x, y = oracle.xgcd(123, P)
    assert g == 1 and (123 * x + P * y) == 1


# --------------------------------------------------------- univariate/*.rs
def test_poly_deg_eq(oracle):
    o = oracle                                            # mod.rs:194-224
    assert o.poly_deg([]) == -1 and o.poly_deg([5]) == 0 and o.poly_deg([1, 2]) == 1
    assert o.poly_deg([1, 2, 3]) == 2 and o.poly_deg([1, 2, 0, 0]) == 1 and o.poly_deg([0, 0]) == -1
    assert o.poly_eq([1, 2], [1, 2]) and not o.poly_eq([1, 2], [1, 3]) and o.poly_eq([1, 2], [1, 2, 0])  # mod.rs:283-302


def test_poly_add_sub_mul(oracle):
    o = oracle
    assert o.poly_add([1, 2], []) == [1, 2] and o.poly_add([], [1, 2]) == [1, 2]
    assert o.poly_add([3], [5]) == [8] and o.poly_add([1, 2], [3, 4]) == [4, 6]
    assert o.poly_add([5], [1, 2]) == [6, 2] == o.poly_add([1, 2], [5])
    assert o.poly_deg(o.poly_add([1, 2], [P - 1, P - 2])) == -1
    assert o.poly_add([P - 1], [2]) == [1]
    assert o.poly_sub([3, 4], []) == [3, 4] and o.poly_sub([], [3, 4]) == [P - 3, P - 4]
    assert o.poly_sub([8], [3]) == [5] and o.poly_sub([6, 8], [2, 3]) == [4, 5]
    assert o.poly_sub([1, 2], [1]) == [0, 2] and o.poly_sub([1], [3]) == [P - 2]
    assert o.poly_deg(o.poly_sub([5, 7], [5, 7])) == -1
    assert o.poly_mul([2, 3], []) == [] and o.poly_mul([], [2, 3]) == []
    assert o.poly_mul([3], [4]) == [12] and o.poly_mul([1, 1], [1, 1]) == [1, 2, 1]
    assert o.poly_mul([2], [1, 0, 1]) == [2, 0, 2]
    assert o.poly_mul([1, 0, 2], [3, 0, 4]) == [3, 0, 10, 0, 8]      # mul.rs sparse
    assert o.poly_mul([P - 1], [2]) == [(2 * (P - 1)) % P]
    assert o.poly_eq(o.poly_mul([1, 2, 3], [4, 5]), o.poly_mul([4, 5], [1, 2, 3]))


def test_poly_div_exp_zerofier(oracle):
    o = oracle
    q, r = o.poly_div([2, 3, 1], [1, 1]);  assert q == [2, 1] and o.poly_deg(r) == -1
    q, r = o.poly_div([1, 0, 1], [1, 1]);  assert o.poly_deg(q) == 1 and o.poly_deg(r) == 0 and r[0] == 2
    q, r = o.poly_div([2, 4, 6], [2]);     assert q == [1, 2, 3] and o.poly_deg(r) == -1
    q, r = o.poly_div([1, 1], [1, 0, 1]);  assert q == [] and r == [1, 1]
    with pytest.raises(o.OraclePanic, match="No division by zero"):
        o.poly_div([1, 1], [])
    q, r = o.poly_div([7, 14], [7]);       assert q == [1, 2]
    q, r = o.poly_div([5, 7, 3, 1], [2, 1])
    assert o.poly_eq(o.poly_add(o.poly_mul(q, [2, 1]), r), [5, 7, 3, 1])
    assert o.poly_exp([1, 2], 0) == [1] and o.poly_exp([3], 4) == [81]
    assert o.poly_exp([1, 1], 2) == [1, 2, 1] and o.poly_exp([1, 1], 3) == [1, 3, 3, 1]
    assert o.poly_exp([], 5) == []
    assert o.poly_deg(o.poly_exp([2, 1], 10)) == 10
    assert o.poly_zerofier([5]) == [P - 5, 1]                        # mod.rs:320-332
    assert o.poly_zerofier([2, 3]) == [6, P - 5, 1]                  # mod.rs:335-350
    assert o.poly_zerofier([1, 2, 3]) == [P - 6, 11, P - 6, 1]       # mod.rs:353-373
    assert o.poly_zerofier([0]) == [0, 1]
    assert o.poly_eval(o.poly_zerofier([1, 2]), 5) == 12             # mod.rs:389-399


def test_poly_scale(oracle):
    o = oracle
    assert o.poly_scale([5], 3) == [5]                               # mod.rs:416-424
    assert o.poly_scale([2, 3], 5) == [2, 15]                        # mod.rs:427-436
    assert o.poly_scale([1, 2, 3], 2) == [1, 4, 12]                  # mod.rs:439-456
    for x in range(1, 6):                                            # mod.rs:459-488
        assert o.poly_eval(o.poly_scale([1, 1, 1], 2), x) == o.poly_eval([1, 1, 1], 2 * x % P)
    dom = [o.ff_exp(2, i) for i in range(4)]                         # mod.rs:491-513
    poly = o.poly_interpolate_domain(dom, [0, 1, 2, 3])
    sc = o.poly_scale(poly, 2)
    for i in range(3):
        assert o.poly_eval(sc, o.ff_exp(2, i)) == i + 1
    assert o.poly_scale([], 5) == []


def test_poly_eval(oracle):
    o = oracle
    assert o.poly_eval([], 5) == 0 and o.poly_eval([7], 10) == 7 and o.poly_eval([2, 3], 4) == 14  # eval.rs:36-63
    assert o.poly_eval([5, 7, 9], 0) == 5 and o.poly_eval([1, 2, 3, 4], 2) == 49                  # eval.rs:66-98
    assert list(o.poly_eval_domain([1, 1], [0, 1, 2, 3])) == [1, 2, 3, 4]                         # eval.rs:101-117
    assert o.poly_eval([P - 1], P - 1) == P - 1                                                   # eval.rs:140-148


def test_poly_interpolate(oracle):
    o = oracle
    assert list(o.poly_interpolate_domain([1, 2, 3], [1, 4, 9])) == [0, 0, 1]          # interpolate.rs:61-84
    assert list(o.poly_interpolate_domain([1, 3], [5, 9])) == [3, 2]                   # interpolate.rs:87-95
    assert list(o.poly_interpolate_domain([1, 2, 3], [2, 5, 10])) == [1, 0, 1]         # interpolate.rs:98-115
    dom, vals = [0, 1, 2, 4], [3, 7, 13, 35]                                           # interpolate.rs:118-136
    c = o.poly_interpolate_domain(dom, vals)
    assert [o.poly_eval(c, x) for x in dom] == vals
    c = o.poly_interpolate_domain([0, 1, P - 5], [P - 2, 6, 48])                       # interpolate.rs:139-163
    assert list(c) == [P - 2, 5, 3]
    # result-shape quirks
    assert len(o.poly_interpolate_domain([1, 2, 3], [0, 0, 0])) == 0
    assert list(o.poly_interpolate_domain([7], [0])) == [0]
    assert list(o.poly_interpolate_domain([1, 2], [5, 5])) == [5, 0]
    with pytest.raises(o.OraclePanic, match="no inverse"):                             # mod.rs:613-625
        o.poly_interpolate_domain([5, 5, 5], [1, 2, 3])


def test_colinearity(oracle):
    o = oracle                                                                          # mod.rs:563-638
    assert o.poly_test_colinearity([(1, 2), (2, 4), (3, 6)])
    assert not o.poly_test_colinearity([(1, 1), (2, 4), (3, 9)])
    assert o.poly_test_colinearity([(5, 7), (10, 99)])
    assert o.poly_test_colinearity([(1, 5), (2, 5), (3, 5)])
    assert o.poly_test_colinearity([(0, 0), (1, 3), (2, 6)])
    with pytest.raises(o.OraclePanic, match="no inverse"):
        o.poly_test_colinearity([(5, 1), (5, 2), (5, 3)])


# ---------------------------------------------------------- hash / merkle
def test_hash_structure(oracle):
    o = oracle                                                                          # hash.rs:106-149
    assert o.hash_from_bytes(b"hello") == o.hash_from_bytes(b"hello")
    assert o.hash_from_bytes(b"hello") != o.hash_from_bytes(b"world")
    h1, h2 = o.hash_from_bytes(b"hello"), o.hash_from_bytes(b"hallo")
    assert sum(a != b for a, b in zip(h1, h2)) > 10
    assert len(o.hash_from_field_elements([1, 2, 3, 4, 5])) == 32
    l, r = o.hash_from_bytes(b"left"), o.hash_from_bytes(b"right")
    assert o.hash_combine(l, r) not in (l, r)
    assert o.hash_from_u64(5) == o.hash_from_field_elements([5]) == o.hash_from_bytes((5).to_bytes(8, "little"))


def test_hash_matches_survey_provisional_digests(oracle):
    """digests minted by the survey's independent Python restatement of
    hash.rs.  NOT reference outputs ("parity unpinned"); two independent restatements
    agreeing is the strongest pin available without a Rust toolchain."""
    o = oracle
    hx = lambda b: b.hex()
    assert hx(o.hash_from_bytes(b"")) == "f2de8d1dbca64572c0310f32459054b28a30a5aa56ade96fa7d71fe77b536a66"
    assert hx(o.hash_from_bytes(b"hello")) == "663afaa74185a1693451aa7fd22ac722ff8f89aabc0471f28dc7c2b7354cae8e"
    h = hx(o.hash_from_bytes(b"hallo"))
    assert h.startswith("45c8705a") and h.endswith("a9253871")
    assert hx(o.hash_from_field_elements([0])) == "3af3b40f826c728865415db948f1befafbe498cea54ea7d7c395b2de4ee7a3df"
    assert hx(o.hash_from_field_elements([5])) == "b41399e39a0d1249b4f0318e5e6416ccaa2dffb01519bf1296173be9c10b5f06"
    assert hx(o.hash_from_field_elements([P - 1])) == "a704c324ff9390ce71ac82dc219c047e1d4168527cc24213cdec6e4efbe406fd"
    assert hx(o.hash_from_field_elements([1, 2, 3, 4, 5])) == "db49061aca4f293c9786bb7785bbeff5ee5d3a4432379873ff027192dfbb8782"
    # leaves as in merkle.rs:104 `Hash::from_bytes(&[i])`
    l4 = np.stack([np.frombuffer(o.hash_from_bytes(bytes([i])), dtype=np.uint8) for i in range(4)])
    assert hx(o.merkle_commit(l4)) == "cefa0c7c9b7c32ab2884c0614d7b1433a783c5190c20ff675a47b8541f122013"
    l8 = np.stack([np.frombuffer(o.hash_from_bytes(bytes([i])), dtype=np.uint8) for i in range(8)])
    assert hx(o.merkle_commit(l8)) == "d86d7c3c1368c029ff23248875ffb2fb673459897e3dcbd67ac0e09ca4cdd738"
    l32 = np.tile(np.frombuffer(o.hash_from_field_elements([5]), dtype=np.uint8), (32, 1))
    assert hx(o.merkle_commit(l32)) == "6ae1c3c0393166bada8c921ab31e658809a416cee02be359efd262077178df70"


def _pyhash(data: bytes) -> bytes:
    """Second, independent restatement of hash.rs:7-99 in pure Python (test-only)."""
    primes = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53]
    rc = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36, 0x6c, 0xd8, 0xab, 0x4d, 0x9a, 0x2f,
          0x5e, 0xbc, 0x63, 0xc6, 0x97, 0x35, 0x6a, 0xd4, 0xb3, 0x7d, 0xfa, 0xef, 0xc5, 0x91, 0x39, 0x72]
    rot = lambda b, n: ((b << n) | (b >> (8 - n))) & 0xFF
    s = [primes[i % 16] for i in range(32)]

    def mix():
        for i in range(32):
            s[i] = rot((s[i] * 251) & 0xFF, 1) ^ 0x63
        for g in range(8):
            t0, t1, t2, t3 = s[4 * g:4 * g + 4]
            s[4 * g:4 * g + 4] = [t0 ^ t1 ^ t3, t0 ^ t2 ^ t3, t0 ^ t1 ^ t2, t1 ^ t2 ^ t3]
        for i in range(32):
            s[i] = (s[i] + s[(i + 1) % 32] + s[31 if i == 0 else i - 1]) & 0xFF
        for i in range(32):
            s[i] = (s[i] + rc[i]) & 0xFF

    for off in range(0, len(data), 32):
        for i, b in enumerate(data[off:off + 32]):
            s[i] = rot((s[i] + b) & 0xFF, 3)
            s[(i + 7) % 32] ^= s[i]
        mix()
    for _ in range(8):
        mix()
    return bytes(s)


def test_hash_two_restatements_agree(oracle):
    rng = np.random.default_rng(7)
    for n in [0, 1, 5, 8, 31, 32, 33, 36, 63, 64, 65, 100, 257]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.hash_from_bytes(data) == _pyhash(data), n


def test_merkle(oracle):
    o = oracle
    leaves = np.stack([np.frombuffer(o.hash_from_bytes(bytes([i])), dtype=np.uint8) for i in range(4)])
    nodes = o.merkle_new(leaves)                                                        # merkle.rs:104-110
    assert len(nodes) == 7
    leaves = np.stack([np.frombuffer(o.hash_from_bytes(bytes([i])), dtype=np.uint8) for i in range(8)])
    nodes = o.merkle_new(leaves)                                                        # merkle.rs:113-122
    root = bytes(nodes[-1])
    for i in range(8):
        path = o.merkle_open(nodes, 8, i)
        assert len(path) == 3
        assert o.merkle_verify(bytes(leaves[i]), i, path, root)
    assert not o.merkle_verify(o.hash_from_bytes(bytes([99])), 0, o.merkle_open(nodes, 8, 0), root)  # merkle.rs:125-133
    with pytest.raises(o.OraclePanic, match="Number of leaves must be power of 2"):
        o.merkle_new(leaves[:3])
    with pytest.raises(o.OraclePanic, match="Cannot create tree from empty leaves"):
        o.merkle_new(leaves[:0])
    with pytest.raises(o.OraclePanic, match="Index out of bounds"):
        o.merkle_open(nodes, 8, 8)


# --------------------------------------------------------------------- fri
def _domain(o, omega, offset, n):
    return [o.ff_mul(offset, o.ff_exp(omega, i)) for i in range(n)]


@pytest.mark.parametrize("n,exp,t,offset,coeffs", [
    (32, 4, 2, 3, [5]),                         # fri.rs:533-563 (constant codeword)
    (64, 4, 3, 7, [5, 3]),                      # fri.rs:566-601
    (128, 4, 4, 13, [1, 3, 2]),                 # fri.rs:604-646
    (256, 8, 5, 17, [1, 2, 5, 3, 7, 4, 1, 2]),  # fri.rs:649-693
])
def test_fri_prove_verify_reference_cases(oracle, n, exp, t, offset, coeffs):
    o = oracle
    omega = o.ff_prim_nth_root(n)
    cfg = o.fri_cfg(omega, offset, n, exp, t)
    codeword = o.poly_eval_domain(coeffs, _domain(o, omega, offset, n))
    proof, top = o.fri_prove(cfg, codeword)
    assert len(top) == t
    assert o.fri_verify(cfg, proof), o.fri_last_reject()
    # the proof must not verify once a revealed value is corrupted
    bad = bytearray(proof)
    bad[-40] ^= 1
    assert not o.fri_verify(cfg, bytes(bad))


def test_fri_num_rounds_and_asserts(oracle):
    o = oracle
    w = o.ff_prim_nth_root
    assert o.fri_num_rounds(o.fri_cfg(w(32), 3, 32, 4, 2)) == 2
    assert o.fri_num_rounds(o.fri_cfg(w(64), 7, 64, 4, 3)) == 3
    assert o.fri_num_rounds(o.fri_cfg(w(128), 13, 128, 4, 4)) == 3
    assert o.fri_num_rounds(o.fri_cfg(w(256), 17, 256, 8, 5)) == 4
    assert o.fri_num_rounds(o.fri_cfg(w(1 << 23), 3, 1 << 23, 8, 32)) == 16
    with pytest.raises(o.OraclePanic, match="Domain length must be power of 2"):
        o.fri_cfg(3, 3, 48, 4, 2)
    with pytest.raises(o.OraclePanic, match="Expansion factor must be power of 2"):
        o.fri_cfg(3, 3, 64, 6, 2)
    with pytest.raises(o.OraclePanic, match="Expansion factor must be at least 4"):
        o.fri_cfg(3, 3, 64, 2, 2)
    cfg = o.fri_cfg(w(32), 3, 32, 4, 2)
    with pytest.raises(o.OraclePanic, match="initial codeword length does not match domain length"):
        o.fri_prove(cfg, [5] * 16)


def test_fri_fold_is_degree_halving(oracle):
    """fold(f)(x^2) = (f(x)+f(-x))/2 + alpha (f(x)-f(-x))/(2x): folding the codeword of
    sum c_j x^j gives the codeword of sum (c_2j + alpha c_2j+1) y^j on the squared domain."""
    o = oracle
    n, offset = 64, 7
    omega = o.ff_prim_nth_root(n)
    cfg = o.fri_cfg(omega, offset, n, 4, 3)
    coeffs = [3, 1, 4, 1, 5, 9, 2, 6]
    cw = o.poly_eval_domain(coeffs, _domain(o, omega, offset, n))
    alpha = 0xDEADBEEFCAFEF00D            # unreduced u64, like FiatShamir::challenge (H6)
    folded = o.fri_fold_codeword(cfg, cw, alpha, offset, omega)
    a = alpha % P
    half = [o.ff_add(coeffs[2 * j], o.ff_mul(a, coeffs[2 * j + 1])) for j in range(4)]
    dom2 = _domain(o, o.ff_mul(omega, omega), o.ff_mul(offset, offset), n // 2)
    assert list(folded) == list(o.poly_eval_domain(half, dom2))
    assert list(o.fast_fold(cw, alpha, offset, omega)) == list(folded)


def test_fri_sampling(oracle):
    o = oracle
    h = bytes(range(32))
    assert o.fri_sample_index(h, 1 << 20) == int.from_bytes(h[24:], "big") % (1 << 20)
    idx = o.fri_sample_indices(b"seed", 64, 16, 8)
    assert len(idx) == 8 and len({i % 16 for i in idx}) == 8 and all(i < 64 for i in idx)
    with pytest.raises(o.OraclePanic):
        o.fri_sample_indices(b"seed", 64, 4, 8)


def test_fiat_shamir(oracle):
    o = oracle
    fs = o.FiatShamir()
    assert fs.challenge() == int.from_bytes(o.hash_from_bytes(b"")[:8], "little")
    fs.absorb(b"abc"); fs.absorb(b"def")
    c1 = fs.challenge()
    assert c1 == fs.challenge() == int.from_bytes(o.hash_from_bytes(b"abcdef")[:8], "little")


def test_trace(oracle):
    import ctypes as C
    o = oracle
    lo = np.zeros(100, dtype=np.uint64); hi = np.zeros(100, dtype=np.uint64)
    o.lib().so_trace_fibonacci(100, lo.ctypes.data_as(o.u64p), hi.ctypes.data_as(o.u64p))
    fib = [1, 1]
    while len(fib) < 100:
        fib.append(fib[-1] + fib[-2])
    assert [int(v) for v in lo] == [f & (2**64 - 1) for f in fib]       # trace.rs:36-49 + `as u64`
    assert int(hi[99]) == fib[99] >> 64
