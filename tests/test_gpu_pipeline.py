"""GPU: device-resident pipeline (LDE -> commit -> combine -> Fri::prove), the four-step pieces
at world size 1, and full-size property checks at BASELINE's sizes.  `pytest -m gpu`."""
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


def _upload(eng, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    d = eng.dev_alloc(arr.size * 4)
    eng.dev_upload(arr, d)
    return d


@pytest.mark.parametrize("which", ["ref_prime", "second_prime"])
def test_stark_prove_composition(eng, eng2, oracle, which):
    """Every stage of the build-defined composition against the oracle: LDE, column roots,
    Fiat-Shamir weights + combination (through the FRI round-0 root), and the oracle's
    Fri::verify on the GPU proof."""
    o = oracle
    e, p, g = (eng, P, G) if which == "ref_prime" else (eng2, P2, G2)
    logn, lb, W, t = 10, 3, 4, 8
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, p) for c in range(W)])
    d = _upload(e, cols)
    res = e.dev_stark_prove(d, W, logn, lb, t, timed=True)
    e.dev_free(d)
    w, Wn = o.ff_prim_nth_root_g(n, p, g), o.ff_prim_nth_root_g(N, p, g)
    lde = [o.fast_coset_ntt(o.fast_intt(cols[c], w, 1, p), N, Wn, g, p) for c in range(W)]
    fs = o.FiatShamir()
    weights = []
    for c in range(W):
        root = o.merkle_commit(o.leaf_hashes(lde[c]))
        assert bytes(res["column_roots"][c]) == root
        fs.absorb(root)
        weights.append(fs.challenge() % p)
    cw = np.zeros(N, dtype=object)
    for c in range(W):
        cw = (cw + lde[c].astype(object) * weights[c]) % p
    cw = cw.astype(np.uint64)
    cfg = o.fri_cfg(Wn, g, N, 1 << lb, t, p)
    want_proof, want_top = o.fri_prove(cfg, cw)
    assert res["top_indices"] == want_top
    assert res["proof"] == want_proof
    assert o.fri_verify(cfg, res["proof"]), o.fri_last_reject()
    assert set(res["stage_ms"]) == {"lde", "commit", "combine", "fri"}


def test_copy_probe_twins_run_and_leave_the_engine_intact(eng, oracle):
    """smi_ctx_copy_probe (bench.py's pattern-copy roofline leg): every pass launches its copy-only
    twin under a name of its own; switching it off restores the transform."""
    o = oracle
    logn, lb, W = 16, 3, 2
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 40 + c, n) for c in range(W)])
    d_in = _upload(eng, cols)
    d_out = eng.dev_alloc(W * N * 4)
    eng.copy_probe(True)
    eng.profile(True)
    eng.dev_lde(d_in, W, logn, lb, d_out, 1, G)
    names = eng.profile_read()
    eng.profile(False)
    eng.copy_probe(False)
    assert names and all(k.startswith("ntt_copy_probe<") for k in names), names
    eng.dev_lde(d_in, W, logn, lb, d_out, 1, G)
    got = eng.dev_download(d_out, W * N).reshape(W, N)
    w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    for c in range(W):
        assert np.array_equal(got[c], o.fast_coset_ntt(o.fast_intt(cols[c], w, 1), N, Wn, G))
    eng.dev_free(d_in)
    eng.dev_free(d_out)


@pytest.mark.parametrize("W", [1, 2, 3, 4, 5, 9])
@pytest.mark.parametrize("logn", [0, 3, 9, 13])
def test_row_leaf_tree_equals_oracle(eng, oracle, W, logn):
    """smi_dev_merkle_build_rows: leaf i = Hash::from_field_elements(row i) (src/hash.rs:32-35 on a
    whole row -- build-defined leaf rule), every level against the oracle's MerkleTree::new."""
    o = oracle
    n = 1 << logn
    cols = np.stack([_vals(o, 90 + 7 * W + c, n) for c in range(W)])
    cols[0, 0] = P - 1
    d_cols = _upload(eng, cols)
    d_nodes = eng.dev_alloc((2 * n - 1) * 32)
    eng.dev_merkle_build_rows(d_cols, W, n, n, d_nodes)
    got = eng.dev_download(d_nodes, (2 * n - 1) * 8).astype(np.uint32).view(np.uint8).reshape(-1, 32)
    leaves = np.stack([np.frombuffer(o.hash_from_field_elements([int(cols[c, i]) for c in range(W)]), dtype=np.uint8) for i in range(n)])
    want = o.merkle_new(leaves)
    assert np.array_equal(got, want)
    eng.dev_free(d_cols)
    eng.dev_free(d_nodes)


def test_stark_prove_row_leaves_variant(eng, oracle):
    """dev_stark_prove with one row-leaf tree instead of a tree per column: the root is the oracle's
    root over the rows of the extended trace, the weights follow the documented transcript
    (root || c), and the oracle's Fri::verify accepts the proof of the combined codeword."""
    o = oracle
    logn, lb, W, t = 10, 3, 4, 8
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n) for c in range(W)])
    d_trace = _upload(eng, cols)
    res = eng.dev_stark_prove(d_trace, W, logn, lb, t, row_leaves=True)
    w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    lde = np.stack([o.fast_coset_ntt(o.fast_intt(cols[c], w, 1), N, Wn, G) for c in range(W)])
    leaves = np.stack([np.frombuffer(o.hash_from_field_elements([int(lde[c, i]) for c in range(W)]), dtype=np.uint8) for i in range(N)])
    root = o.merkle_commit(leaves)
    assert res["column_roots"].shape == (1, 32) and bytes(res["column_roots"][0]) == root
    weights = [int.from_bytes(o.hash_from_bytes(root + c.to_bytes(8, "little"))[:8], "little") % P for c in range(W)]
    cw = np.zeros(N, dtype=object)
    for c in range(W):
        cw = (cw + weights[c] * lde[c].astype(object)) % P
    cfg = o.fri_cfg(Wn, G, N, 1 << lb, t)
    want, want_top = o.fri_prove(cfg, np.array(cw, dtype=np.uint64))
    assert res["proof"] == want and res["top_indices"] == want_top and o.fri_verify(cfg, res["proof"])
    eng.dev_free(d_trace)


def test_cfg3_full_size_lde_and_commit_properties(eng, oracle):
    """BASELINE configs[2]: 2^20-row x 4-column trace, blowup 8 (N = 2^23, the largest domain the
    reference prime has) + Merkle commit.  Size-independent properties: the extension agrees with
    the trace on the subgroup it extends, the op-for-op oracle's Polynomial::eval at sampled
    points, interpolating the extension back gives a degree < n polynomial, and sampled Merkle
    paths verify with the oracle's MerkleTree::verify."""
    o = oracle
    logn, lb, W = 20, 3, 4
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n) for c in range(W)])
    d_in = _upload(eng, cols)
    d_out = eng.dev_alloc(W * N * 4)
    eng.dev_lde(d_in, W, logn, lb, d_out, 1, G)
    lde = eng.dev_download(d_out, W * N).reshape(W, N)
    w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    for c in (0, 3):
        coeffs = eng.intt(cols[c], 1)
        for k in (0, 1, 777777, N - 1):                                      # oracle eval.rs:6-14 at x_k
            assert o.poly_eval(coeffs, o.ff_mul(G, o.ff_exp(Wn, k))) == int(lde[c, k])
        back = eng.intt(lde[c], G)                                           # degree < n: high coefficients vanish
        assert np.array_equal(back[:n], coeffs) and not back[n:].any()
    # the same extension on offset 1 contains the trace itself at stride 8
    eng.dev_lde(d_in, W, logn, lb, d_out, 1, 1)
    plain = eng.dev_download(d_out, N)
    assert np.array_equal(plain[::8], cols[0])
    # commit column 0 of the coset extension; sampled openings verified by the oracle
    d_nodes = eng.dev_alloc((2 * N - 1) * 32)
    eng.dev_lde(d_in, W, logn, lb, d_out, 1, G)
    eng.dev_merkle_build(d_out, N, d_nodes)
    tree = eng.merkle_from_codeword(lde[0])
    root = tree.root()
    for i in (0, 1, 4242424, N - 1):
        assert o.merkle_verify(o.hash_from_field_elements([int(lde[0, i])]), i, tree.open(i), root)
    # root produced by the device-resident build equals the host-buffer build
    words = eng.dev_download(d_nodes + (2 * N - 2) * 32, 8)
    assert b"".join(int(v).to_bytes(4, "little") for v in words) == root
    tree.free()
    for ptr in (d_in, d_out, d_nodes):
        eng.dev_free(ptr)


def test_full_size_fri_prove_accepted_by_oracle_verify(eng, oracle):
    """The parity-checked twin of BASELINE configs[4] on the reference prime: a 2^23-point
    codeword of a degree < 2^20 polynomial, expansion 8, 32 colinearity tests -> 16 rounds, last
    codeword 256.  The oracle's Fri::verify (src/fri.rs:313-504) must accept the GPU proof, and
    reject it after a one-bit corruption."""
    o = oracle
    logn, lb, t = 20, 3, 32
    N = 1 << (logn + lb)
    Wn = o.ff_prim_nth_root(N)
    coeffs = _vals(o, 99, 1 << logn)
    codeword = eng.coset_ntt(coeffs, logn + lb, G)
    cfg = eng.fri_cfg(Wn, G, N, 1 << lb, t)
    assert eng.fri_num_rounds(cfg) == 16
    proof, top = eng.fri_prove(cfg, codeword)
    ocfg = o.fri_cfg(Wn, G, N, 1 << lb, t)
    ok, values = o.fri_verify(ocfg, proof, want_values=True)
    assert ok, o.fri_last_reject()
    assert len(values) == 2 * t and all(int(codeword[i]) == v for i, v in values)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    assert not o.fri_verify(ocfg, bytes(bad))


def test_fri_prove_with_folds_computed_by_the_leaf_kernel_is_byte_identical(eng, oracle):
    """From 2^20 leaves on, a round's tree kernel computes its own leaves: Fri::fold_codeword of the previous codeword (LeafSrc /
    LEAF_FOLD, csrc/hash.hip) instead of reading the output of a separate fold launch.  2^21-point codeword: round 1 (2^20
    leaves) takes that path -- the serialized proof must still be the op-for-op oracle's, byte for byte, and the retained
    codeword of round 1 the oracle's fold of round 0 with the oracle's challenge."""
    o = oracle
    n, exp, t, offset = 1 << 21, 8, 8, 5
    omega = o.ff_prim_nth_root(n)
    cw = o.fast_coset_ntt(_vals(o, 77, n // exp), n, omega, offset)
    cfg_o, cfg_e = o.fri_cfg(omega, offset, n, exp, t), eng.fri_cfg(omega, offset, n, exp, t)
    want, want_top = o.fri_prove(cfg_o, cw)
    d = _upload(eng, cw)
    got, got_top = eng.dev_fri_prove(cfg_e, d, n)
    eng.dev_free(d)
    assert list(got_top) == want_top
    assert bytes(got) == want
    # a caller's codeword that is only 4-byte aligned: the four-at-a-time leaf source must not be used on it (separate fold launch)
    d2 = eng.dev_alloc((n + 4) * 4)
    eng.dev_upload(cw, d2 + 4)
    got2, top2 = eng.dev_fri_prove(cfg_e, d2 + 4, n)
    eng.dev_free(d2)
    assert bytes(got2) == want and list(top2) == want_top


def test_stark_prove_with_the_combination_computed_by_the_leaf_kernel(eng2, oracle):
    """2^17 rows x 4 columns at blowup 8 (N = 2^20): the first FRI tree's kernel computes the weighted column sum itself
    (LEAF_COMBINE) -- the proof must equal the one made step by step through the C ABI with the stand-alone
    smi_dev_combine_columns (whose own parity with the oracle test_stark_prove_composition checks), and the oracle's
    Fri::verify must accept it."""
    import ctypes as C
    o = oracle
    e, p, g = eng2, P2, G2
    logn, lb, W, t = 17, 3, 4, 8
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, p) for c in range(W)])
    d = _upload(e, cols)
    res = e.dev_stark_prove(d, W, logn, lb, t)
    # step by step: extension, column roots (Merkle over each column), weights on the host from the roots, combination, Fri::prove
    d_lde = e.dev_alloc(W * N * 4)
    e.dev_lde(d, W, logn, lb, d_lde)
    d_nodes = e.dev_alloc((2 * N - 1) * 32)
    fs, weights = o.FiatShamir(), []
    for c in range(W):
        e.dev_merkle_build(d_lde + c * N * 4, N, d_nodes)
        e.sync()
        root = e.dev_download(d_nodes + (2 * N - 2) * 32, 8).astype(np.uint32).tobytes()     # eight LE words = the 32 root bytes
        assert root == bytes(res["column_roots"][c])
        fs.absorb(root)
        weights.append(fs.challenge())                      # unreduced, as the device keeps them
    import torch
    t_w = torch.from_numpy(np.array(weights, dtype=np.uint64).view(np.int64)).cuda()
    d_cw = e.dev_alloc(N * 4)
    e.dev_combine_columns(d_lde, W, N, N, t_w.data_ptr(), d_cw)
    Wn = e.prim_nth_root(N)
    want, want_top = e.dev_fri_prove(e.fri_cfg(Wn, g, N, 1 << lb, t), d_cw, N)
    assert res["proof"] == bytes(want) and res["top_indices"] == list(want_top)
    assert o.fri_verify(o.fri_cfg(Wn, g, N, 1 << lb, t, p), res["proof"]), o.fri_last_reject()
    e.sync()
    for ptr in (d, d_lde, d_nodes, d_cw):
        e.dev_free(ptr)


def test_mix_probe_and_the_librarys_mix_accounting(eng, oracle):
    """bench.py's prove_roofline: smi_ctx_mix_probe (the bare permutation's rate, the in-run ceiling) returns a plausible rate,
    and the per-kernel records count SURVEY 8(d)'s N*9 + (N-1)*10 mix_state evaluations for a tree -- whichever kernels the
    planner splits it into (chunk kernel only at 2^12, subtree + chunk kernels at 2^20 and 2^21)."""
    rate = eng.mix_probe(64)
    assert 5e10 < rate < 5e12
    for logn in (12, 20, 21):
        n = 1 << logn
        d = _upload(eng, _vals(oracle, logn, n))
        d_nodes = eng.dev_alloc((2 * n - 1) * 32)
        eng.profile(True)
        eng.dev_merkle_build(d, n, d_nodes)
        k = eng.profile_read()
        eng.profile(False)
        assert sum(v["alg_mixes"] for v in k.values()) == 9.0 * n + 10.0 * (n - 1), (logn, k)
        eng.dev_free(d)
        eng.dev_free(d_nodes)


def test_cfg5_shape_on_second_prime(eng2, oracle):
    """BASELINE configs[4] shape (2^22 rows x 4 columns, blowup 8 -> 2^25 domain, 18 rounds) on the
    second prime; accepted by the p-generic oracle's Fri::verify."""
    o = oracle
    logn, lb, W, t = 22, 3, 4, 32
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, P2) for c in range(W)])
    d = _upload(eng2, cols)
    res = eng2.dev_stark_prove(d, W, logn, lb, t, timed=True)
    eng2.dev_free(d)
    Wn = o.ff_prim_nth_root_g(N, P2, G2)
    cfg = o.fri_cfg(Wn, G2, N, 1 << lb, t, P2)
    assert o.fri_num_rounds(cfg) == 18
    assert o.fri_verify(cfg, res["proof"]), o.fri_last_reject()


# ---- two-pass low-degree extension (csrc/lde_core.h): smi_dev_lde at the sizes it serves
@pytest.mark.parametrize("which,logn,lb,W", [("ref", 20, 1, 1), ("ref", 20, 2, 2), ("ref", 20, 3, 4), ("ref", 21, 2, 1),
                                             ("second", 20, 4, 1), ("second", 21, 3, 2), ("second", 22, 3, 4), ("second", 22, 4, 1)])
def test_two_pass_lde_equals_oracle(eng, eng2, oracle, which, logn, lb, W):
    """interpolate_domain on the trace domain, eval_domain on the blowup coset (src/univariate/
    interpolate.rs:6-44, eval.rs:16-21) -- the oracle's fast restatement of both, whole columns."""
    o = oracle
    e, p, g = (eng, P, G) if which == "ref" else (eng2, P2, G2)
    n, N = 1 << logn, 1 << (logn + lb)
    w, Wn = o.ff_prim_nth_root_g(n, p, g), o.ff_prim_nth_root_g(N, p, g)
    cols = np.stack([_vals(o, 900 + 7 * logn + c, n, p) for c in range(W)])
    if W > 1:
        cols[1, :] = p - 1                       # lazy sums at their bounds
    d_in = _upload(e, cols)
    d_out = e.dev_alloc(W * N * 4)
    e.lde_two_pass(True)
    e.profile(True)
    e.dev_lde(d_in, W, logn, lb, d_out, 1, g)
    names = e.profile_read()
    e.profile(False)
    e.lde_two_pass(False)
    assert any(k.startswith("lde_a_kernel") for k in names) and "lde_b_kernel" in names, names   # the two-pass path ran
    got = e.dev_download(d_out, W * N).reshape(W, N)
    for c in range(W):
        assert np.array_equal(got[c], o.fast_coset_ntt(o.fast_intt(cols[c], w, 1, p), N, Wn, g, p)), c
    e.dev_free(d_in)
    e.dev_free(d_out)


@pytest.mark.parametrize("which,logn,lb,W", [("ref", 19, 3, 6), ("second", 20, 3, 2), ("second", 22, 1, 5)])
def test_lde_with_the_columns_of_a_tile_in_one_workgroup(eng, eng2, oracle, which, logn, lb, W):
    """ntt_pass_cols_kernel (csrc/ntt.hip): middle passes and the inverse transforms' scaled last passes take up to
    four columns of a tile per workgroup and derive each thread's 16 output multipliers once.  Whole columns
    against the oracle (interpolate_domain + eval_domain restated, src/univariate/interpolate.rs:6-44,
    eval.rs:16-21), with one and with two column groups, and the launch seen in the profile."""
    o = oracle
    e, p, g = (eng, P, G) if which == "ref" else (eng2, P2, G2)
    n, N = 1 << logn, 1 << (logn + lb)
    w, Wn = o.ff_prim_nth_root_g(n, p, g), o.ff_prim_nth_root_g(N, p, g)
    cols = np.stack([_vals(o, 1300 + 5 * logn + c, n, p) for c in range(W)])
    cols[W - 1, :] = p - 1                       # lazy sums at their bounds
    d_in = _upload(e, cols)
    d_out = e.dev_alloc(W * N * 4)
    e.profile(True)
    e.dev_lde(d_in, W, logn, lb, d_out, 1, g)
    names = e.profile_read()
    e.profile(False)
    assert any(k.startswith("ntt_pass_cols_kernel") and k.endswith("mid>") for k in names), names
    if logn >= 22:   # enough tiles for the inverse transforms' scaled last pass to share as well (NttPass::share_cols)
        assert any(k.startswith("ntt_pass_cols_kernel") and k.endswith("last>") for k in names), names
    got = e.dev_download(d_out, W * N).reshape(W, N)
    for c in range(W):
        assert np.array_equal(got[c], o.fast_coset_ntt(o.fast_intt(cols[c], w, 1, p), N, Wn, g, p)), c
    e.dev_free(d_in)
    e.dev_free(d_out)


def test_profile_only_brackets_the_named_kernel_alone(eng2, oracle):
    """smi_ctx_profile_only: bench.py times its dominant kernel with only that kernel bracketed"""
    o = oracle
    logn, lb, W = 18, 3, 2
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 77 + c, n, P2) for c in range(W)])
    d_in = _upload(eng2, cols)
    d_out = eng2.dev_alloc(W * N * 4)
    eng2.profile(True)
    eng2.dev_lde(d_in, W, logn, lb, d_out, 1, G2)
    every = eng2.profile_read()
    assert len(every) >= 2
    one = sorted(every)[0]
    eng2.profile_only(one)
    eng2.dev_lde(d_in, W, logn, lb, d_out, 1, G2)
    only = eng2.profile_read()
    assert list(only) == [one] and only[one]["launches"] == every[one]["launches"]
    eng2.profile_only(None)
    eng2.dev_lde(d_in, W, logn, lb, d_out, 1, G2)
    assert sorted(eng2.profile_read()) == sorted(every)
    eng2.profile(False)
    with pytest.raises(Exception):
        eng2.profile_only("x" * 80)
    eng2.dev_free(d_in)
    eng2.dev_free(d_out)


@pytest.mark.parametrize("which", ["ref_prime", "second_prime"])
def test_stark_prove_column_openings(eng, eng2, oracle, which):
    """smi_stark_cfg.open_columns: the FRI proof bytes are unchanged, the appended openings are byte for byte
    the oracle's restatement (tests/conftest.py), the host mirror's verifier accepts them, and tampering with
    a column value or swapping a root is rejected."""
    from conftest import column_openings_bytes
    from stark_rs_amd.mirror import verify_column_openings
    o = oracle
    e, p, g = (eng, P, G) if which == "ref_prime" else (eng2, P2, G2)
    logn, lb, W, t = 10, 3, 4, 8
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, p) for c in range(W)])
    d = _upload(e, cols)
    plain = e.dev_stark_prove(d, W, logn, lb, t)
    res = e.dev_stark_prove(d, W, logn, lb, t, open_columns=True)
    e.dev_free(d)
    assert res["proof"][:len(plain["proof"])] == plain["proof"] and res["top_indices"] == plain["top_indices"]
    w, Wn = o.ff_prim_nth_root_g(n, p, g), o.ff_prim_nth_root_g(N, p, g)
    lde = [o.fast_coset_ntt(o.fast_intt(cols[c], w, 1, p), N, Wn, g, p) for c in range(W)]
    assert res["proof"][len(plain["proof"]):] == column_openings_bytes(o, lde, res["top_indices"], N)
    roots = [bytes(r) for r in res["column_roots"]]
    assert verify_column_openings(e, res["proof"], W, logn, lb, t, roots, res["top_indices"])
    bad = bytearray(res["proof"])
    bad[len(plain["proof"]) + 9] ^= 1                                  # col_0[a] of test 0
    assert not verify_column_openings(e, bytes(bad), W, logn, lb, t, roots, res["top_indices"])
    assert not verify_column_openings(e, res["proof"], W, logn, lb, t, roots[1:] + roots[:1], res["top_indices"])
