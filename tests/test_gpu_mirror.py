"""GPU: the reference's own unit tests for the hot path, transliterated onto the host-side mirror
(stark_rs_amd/mirror.py), whose heavy methods run on the device through the C ABI.  Each test
names the reference test it follows.  `pytest -m gpu`."""
import pytest

pytestmark = pytest.mark.gpu

P = 998244353


@pytest.fixture(scope="module")
def m():
    import stark_rs_amd.mirror as mirror
    mirror.default_engine()   # fails loudly without libstarkmi.so + GPU
    return mirror


@pytest.fixture(scope="module")
def field(m):
    return m.FiniteField(P)


def fe(field, *vals):
    return [field.new_element(v) for v in vals]


# ---- src/ff.rs tests (host glue; kept so the mirror's semantics are pinned too)
def test_ff_kats(m, field):
    f = field
    assert f.add(*fe(f, P - 1, 5)).value == 4                      # test_addition_modular
    assert f.sub(*fe(f, 5, 10)).value == P - 5                     # test_subtraction_underflow
    assert f.mul(*fe(f, 1000000, 2000000)).value == 2000000000000 % P
    assert f.neg(f.new_element(100)).value == P - 100 and f.neg(f.zero()) == f.zero()
    for a in (123, 1, P - 1):
        x = f.new_element(a)
        assert f.mul(x, f.inv(x)) == f.one()
    with pytest.raises(m.PANIC, match="no inverse"):
        f.inv(f.zero())
    with pytest.raises(m.PANIC, match="no division by zero"):
        f.div(f.new_element(100), f.zero())
    assert f.exp(f.new_element(2), 10).value == 1024 and f.g().value == 3
    assert (f.new_element(3) ^ 2) == f.new_element(9)
    for i in range(1, 11):                                         # test_prim_nth_root_powers_of_two
        assert f.exp(f.prim_nth_root(1 << i), 1 << i) == f.one()
    with pytest.raises(m.PANIC, match="n must be a power of two"):
        f.prim_nth_root(6)
    with pytest.raises(m.PANIC, match="n > 2\\^23 not supported"):
        f.prim_nth_root(1 << 24)
    with pytest.raises(m.PANIC):
        m.FiniteField(2147483647).g()
    assert f.sample(b"").value == 0 and f.sample(bytes([42])).value == 42


# ---- src/univariate/interpolate.rs tests, on geometric domains (the fast-path contract)
def test_interpolate_matches_values_on_subgroup(m, field):
    f = field
    n = 8
    w = f.prim_nth_root(n)
    domain = [f.exp(w, k) for k in range(n)]
    values = fe(f, 3, 7, 13, 35, 1, 0, P - 1, 99)
    poly = m.Polynomial.interpolate_domain(domain, values)        # test_interpolation_matches_values
    assert len(poly.coeffs) == n
    for x, want in zip(domain, values):
        assert poly.eval(x) == want


def test_interpolate_known_polynomials(m, field):
    f = field
    w = f.prim_nth_root(4)
    offset = f.new_element(3)
    domain = [offset * f.exp(w, k) for k in range(4)]
    # x^2 through its own evaluations (test_x2), 2x + 3 (test_linear_polynomial), x^2 + 1 (test_quadratic_polynomial)
    for coeffs in ([0, 0, 1], [3, 2], [1, 0, 1], [P - 2, 5, 3]):
        poly = m.Polynomial(fe(f, *coeffs), f)
        values = [poly.eval(x) for x in domain]
        got = m.Polynomial.interpolate_domain(domain, values)
        assert got == poly                                          # equality ignores trailing zeros (mod.rs:13-39)
        assert len(got.coeffs) == 4
        assert [c.value for c in got.coeffs[:len(coeffs)]] == coeffs
    # all-zero values: the reference returns the empty polynomial (H8) ...
    assert m.Polynomial.interpolate_domain(domain, fe(f, 0, 0, 0, 0)).coeffs == []
    # ... except on a one-point domain, where the single Lagrange term survives Polynomial::add's
    # trimming: interpolate_domain([x], [0]) is [0], not [] (interpolate.rs:21-42 with add.rs:7-12) -- and a
    # one-point domain with a non-zero value gives that constant
    one = m.Polynomial.interpolate_domain(fe(f, 3), fe(f, 0))
    assert [c.value for c in one.coeffs] == [0]
    assert [c.value for c in m.Polynomial.interpolate_domain(fe(f, 3), fe(f, 7)).coeffs] == [7]
    with pytest.raises(m.PANIC, match="no inverse"):               # duplicate points (mod.rs:613-625)
        m.Polynomial.interpolate_domain(fe(f, 5, 5, 5, 5), fe(f, 1, 2, 3, 4))
    with pytest.raises(m.PANIC, match="not offset"):               # outside the fast-path contract
        m.Polynomial.interpolate_domain(fe(f, 1, 2, 3, 4), fe(f, 1, 4, 9, 16))


# ---- src/univariate/eval.rs tests
def test_eval_kats(m, field):
    f = field
    assert m.Polynomial.zero_poly(f).eval(f.new_element(5)).value == 0          # test_eval_zero_poly
    assert m.Polynomial.constant_poly(f, 7).eval(f.new_element(10)).value == 7  # test_eval_constant
    assert m.Polynomial.linear_poly(f, 2, 3).eval(f.new_element(4)).value == 14
    assert m.Polynomial(fe(f, 1, 2, 3, 4), f).eval(f.new_element(2)).value == 49


def test_eval_domain_in_domain_order(m, field):
    f = field
    poly = m.Polynomial(fe(f, 1, 2, 5, 3, 7, 4, 1, 2), f)
    n = 64
    w = f.prim_nth_root(n)
    offset = f.new_element(17)
    domain = [offset * f.exp(w, k) for k in range(n)]
    got = poly.eval_domain(domain)                                   # test_eval_domain, on a coset
    assert len(got) == n
    for x, y in zip(domain, got):
        assert poly.eval(x) == y


# ---- src/univariate/mod.rs scale tests
def test_scale(m, field):
    f = field
    assert [c.value for c in m.Polynomial.constant_poly(f, 5).scale(f.new_element(3)).coeffs] == [5]
    assert [c.value for c in m.Polynomial.linear_poly(f, 2, 3).scale(f.new_element(5)).coeffs] == [2, 15]
    assert [c.value for c in m.Polynomial(fe(f, 1, 2, 3), f).scale(f.new_element(2)).coeffs] == [1, 4, 12]
    poly = m.Polynomial(fe(f, 1, 1, 1), f)
    c = f.new_element(2)
    for t in range(1, 6):                                            # test_scale_evaluation_shift
        x = f.new_element(t)
        assert poly.scale(c).eval(x) == poly.eval(c * x)
    assert m.Polynomial(fe(f, 1, 2, 3), f).scale(f.one()) == m.Polynomial(fe(f, 1, 2, 3), f)
    assert m.Polynomial.zero_poly(f).scale(f.new_element(5)).is_zero()


# ---- src/hash.rs tests
def test_div_and_zerofier(m, field, oracle):
    """src/univariate/div.rs:83-262 and mod.rs:320-413 on the mirror (device NTT products)."""
    f, Poly = field, m.Polynomial
    q, r = Poly(fe(f, 2, 3, 1), f) / Poly(fe(f, 1, 1), f)               # test_division_basic
    assert q.deg() == 1 and [c.value for c in q.coeffs] == [2, 1] and r.is_zero()
    q, r = Poly.div(Poly(fe(f, 1, 0, 1), f), Poly(fe(f, 1, 1), f))       # test_division_with_remainder
    assert q.deg() == 1 and r.deg() == 0 and r.coeffs[0].value == 2
    dividend = Poly(fe(f, 1, 1), f)                                      # test_division_lower_degree_dividend
    q, r = Poly.div(dividend, Poly(fe(f, 1, 0, 1), f))
    assert q.is_zero() and r == dividend
    with pytest.raises(m.PANIC, match="No division by zero"):            # test_division_by_zero
        Poly.div(dividend, Poly([], f))
    dividend = Poly(fe(f, 6, 11, 6, 1), f)                               # test_division_verification
    divisor = Poly(fe(f, 2, 1), f)
    q, r = Poly.div(dividend, divisor)
    back = q * divisor
    back = Poly([f.add(c, r.coeffs[i]) if i < len(r.coeffs) else c for i, c in enumerate(back.coeffs)], f)
    assert back == dividend
    assert Poly.intdiv(dividend, divisor) == q and Poly.modulo(dividend, divisor).is_zero()
    z = Poly.zerofier(fe(f, 5))                                          # test_zerofier_single_point
    assert z.deg() == 1 and [c.value for c in z.coeffs] == [P - 5, 1] and z.eval(f.new_element(5)).value == 0
    z = Poly.zerofier(fe(f, 2, 3))                                       # test_zerofier_two_points
    assert [c.value for c in z.coeffs] == [6, P - 5, 1]
    z = Poly.zerofier(fe(f, 1, 2, 3))                                    # test_zerofier_three_points
    assert [c.value for c in z.coeffs] == [P - 6, 11, P - 6, 1]
    assert [c.value for c in Poly.zerofier(fe(f, 0)).coeffs] == [0, 1]   # test_zerofier_zero_point
    assert Poly.zerofier(fe(f, 1, 2)).eval(f.new_element(5)).value == 12  # test_zerofier_nonzero_evaluation
    for n in (1, 2, 5, 10, 33):                                          # test_zerofier_degree (+ oracle values)
        dom = list(range(1, n + 1))
        z = Poly.zerofier(fe(f, *dom))
        assert z.deg() == n and [c.value for c in z.coeffs] == oracle.poly_zerofier(dom)
    # a zerofier divides what vanishes on its domain
    zs = Poly.zerofier(fe(f, 3, 9, 27, 81))
    g = Poly(fe(f, 7, 0, 5, 1, 2), f)
    assert Poly.modulo(g * zs, zs).is_zero() and Poly.intdiv(g * zs, zs) == g


def test_hash(m):
    H = m.Hash
    assert H.from_bytes(b"hello") == H.from_bytes(b"hello")          # test_hash_deterministic
    assert H.from_bytes(b"hello") != H.from_bytes(b"world")          # test_hash_different_inputs
    h1, h2 = H.from_bytes(b"hello"), H.from_bytes(b"hallo")          # test_hash_avalanche_effect
    assert sum(a != b for a, b in zip(h1.bytes, h2.bytes)) > 10
    assert len(H.from_field_elements([1, 2, 3, 4, 5]).bytes) == 32   # test_hash_from_field_elements
    l, r = H.from_bytes(b"left"), H.from_bytes(b"right")             # test_hash_combine
    c = H.combine(l, r)
    assert c != l and c != r
    assert H.from_u64(5) == H.from_field_elements([5])


# ---- src/merkle.rs tests
def test_merkle(m):
    H, T = m.Hash, m.MerkleTree
    leaves = [H.from_bytes(bytes([i])) for i in range(4)]
    tree = T.new(leaves)                                             # test_merkle_tree_creation
    assert len(tree.leaves) == 4 and len(tree.nodes) == 3
    leaves = [H.from_bytes(bytes([i])) for i in range(8)]
    tree = T.new(leaves)                                             # test_merkle_proof_verification
    for i in range(8):
        assert T.verify(leaves[i], i, tree.open(i), tree.get_root())
    assert T.commit(leaves) == tree.get_root()
    leaves = leaves[:4]
    tree = T.new(leaves)                                             # test_merkle_proof_invalid
    assert not T.verify(H.from_bytes(bytes([99])), 0, tree.open(0), tree.get_root())
    with pytest.raises(m.PANIC, match="Number of leaves must be power of 2"):
        T.new(leaves[:3])
    with pytest.raises(m.PANIC, match="Cannot create tree from empty leaves"):
        T.new([])
    with pytest.raises(m.PANIC, match="Index out of bounds"):
        tree.open(4)


# ---- src/fri.rs tests: prove -> serialize -> deserialize -> verify
@pytest.mark.parametrize("n,expansion,t,offset,coeffs", [
    (32, 4, 2, 3, [5]),                         # test_fri_prove_verify_constant
    (64, 4, 3, 7, [5, 3]),                      # test_fri_prove_verify_linear_polynomial
    (128, 4, 4, 13, [1, 3, 2]),                 # test_fri_prove_verify_quadratic_polynomial
    (256, 8, 5, 17, [1, 2, 5, 3, 7, 4, 1, 2]),  # test_fri_prove_verify_high_degree_polynomial
])
def test_fri_prove_verify(m, field, oracle, n, expansion, t, offset, coeffs):
    f = field
    omega = f.prim_nth_root(n)
    off = f.new_element(offset)
    fri = m.Fri.new(omega, off, n, expansion, t)
    poly = m.Polynomial.new(fe(f, *coeffs), f)
    domain = [f.mul(off, f.exp(omega, i)) for i in range(n)]
    codeword = poly.eval_domain(domain) if len(coeffs) > 1 else [f.new_element(coeffs[0])] * n
    proof_stream = m.ProofStream.new()
    prover_fiat_shamir = m.FiatShamir.new()
    top = fri.prove(list(codeword), prover_fiat_shamir, proof_stream)
    assert len(top) == t
    proof_bytes = proof_stream.serialize()
    # the reference test's own ending: deserialize, then Fri::verify with a fresh FiatShamir
    verifier_stream = m.ProofStream.deserialize(proof_bytes, f)
    verifier_fiat_shamir = m.FiatShamir.new()
    polynomial_values = []
    assert fri.verify(verifier_stream, verifier_fiat_shamir, polynomial_values), "FRI verification should succeed"
    assert len(polynomial_values) == 2 * t and all(codeword[i] == v for i, v in polynomial_values)
    bad = bytearray(proof_bytes)
    bad[-40] ^= 1
    assert not fri.verify(m.ProofStream.deserialize(bytes(bad), f), m.FiatShamir.new(), [])
    # ... and the oracle's restatement of Fri::verify agrees
    assert m.ProofStream.deserialize(proof_bytes, f).serialize() == proof_bytes
    ocfg = oracle.fri_cfg(omega.value, offset, n, expansion, t)
    assert oracle.fri_verify(ocfg, proof_bytes), oracle.fri_last_reject()
    # the prover's transcript now holds the roots, like the reference's (fri.rs:131)
    assert len(prover_fiat_shamir.transcript) == 32 * fri.num_rounds()


def test_fri_new_asserts(m, field):
    f = field
    w = f.prim_nth_root(64)
    with pytest.raises(m.PANIC, match="Domain length must be power of 2"):
        m.Fri.new(w, f.new_element(3), 48, 4, 2)
    with pytest.raises(m.PANIC, match="Expansion factor must be power of 2"):
        m.Fri.new(w, f.new_element(3), 64, 6, 2)
    with pytest.raises(m.PANIC, match="Expansion factor must be at least 4"):
        m.Fri.new(w, f.new_element(3), 64, 2, 2)
    fri = m.Fri.new(f.prim_nth_root(32), f.new_element(3), 32, 4, 2)
    with pytest.raises(m.PANIC, match="initial codeword length does not match domain length"):
        fri.prove([f.new_element(5)] * 16, m.FiatShamir.new(), m.ProofStream.new())


def test_fiat_shamir_and_trace(m, field, oracle):
    fs = m.FiatShamir.new()
    fs.absorb(b"abc")
    fs.absorb(b"def")
    assert fs.challenge(field).value == int.from_bytes(oracle.hash_from_bytes(b"abcdef")[:8], "little")
    tr = m.Trace.fibonacci(64)                                       # trace.rs:36-49
    assert tr.get_col(0)[:6] == [1, 1, 2, 3, 5, 8] and tr.num_columns == 1
    ext = tr.lde(field, 3)
    assert ext.shape == (1, 512)
    col = [v % P for v in tr.get_col(0)]
    # the extension on offset 1 reproduces the trace at stride 8; on the coset it agrees with eval
    plain = tr.lde(field, 3, lde_offset=1)
    assert [int(v) for v in plain[0][::8]] == col
