"""GPU: Fri::verify behind the C ABI (smi_fri_verify, csrc/verify.hip) against the oracle's op-for-op
restatement of src/fri.rs:313-504 -- same verdict, same polynomial_values, on the reference's own four
accept cases (src/fri.rs:533-693), on larger proofs, and on hundreds of tampered proofs (every kind of
object: roots, last codeword, triples, paths, tags, lengths, truncation); and the verifier of the
build-defined composition (smi_stark_verify) with and without column openings.  `pytest -m gpu`."""
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


def _proof(o, n, exp, t, offset, seed, p=P, g=G):
    omega = o.ff_prim_nth_root_g(n, p, g)
    cw = o.fast_coset_ntt(o.splitmix64(seed, n // exp) % np.uint64(p), n, omega, offset, p)
    cfg = o.fri_cfg(omega, offset, n, exp, t, p)
    proof, _top = o.fri_prove(cfg, cw)
    return omega, cfg, proof


def _agree(o, eng, ocfg, ecfg, proof):
    """same verdict and values as the oracle; a reference panic must surface as an exception on both sides"""
    try:
        want = o.fri_verify(ocfg, proof, want_values=True)
        panic = None
    except Exception as e:          # the oracle raises on a reference panic
        want, panic = None, str(e)
    if panic is not None:
        with pytest.raises(Exception):
            eng.fri_verify(ecfg, proof)
        return "panic"
    ok, pv, why = eng.fri_verify(ecfg, proof)
    assert ok == want[0], (why, o.fri_last_reject())
    assert pv == want[1]
    if not ok:
        assert why == o.fri_last_reject()
    return ok


@pytest.mark.parametrize("n,exp,t,offset", [(32, 4, 2, 3), (64, 4, 3, 7), (128, 4, 4, 13), (256, 8, 5, 17), (1 << 14, 8, 16, 3)])
def test_fri_verify_accepts_what_the_reference_accepts(eng, oracle, n, exp, t, offset):
    o = oracle
    omega, ocfg, proof = _proof(o, n, exp, t, offset, 11 + n)
    ecfg = eng.fri_cfg(omega, offset, n, exp, t)
    assert _agree(o, eng, ocfg, ecfg, proof) is True
    # the proof the device prover writes is the same bytes, so it is accepted as well
    cw = o.fast_coset_ntt(o.splitmix64(11 + n, n // exp) % np.uint64(P), n, omega, offset)
    got, _ = eng.fri_prove(ecfg, cw)
    assert bytes(got) == proof


def test_fri_verify_rejects_exactly_what_the_reference_rejects(eng, oracle):
    o = oracle
    n, exp, t, offset = 256, 8, 5, 17
    omega, ocfg, proof = _proof(o, n, exp, t, offset, 5)
    ecfg = eng.fri_cfg(omega, offset, n, exp, t)
    rng = np.random.default_rng(7)
    verdicts = {True: 0, False: 0, "panic": 0}
    # single byte flips at random positions (roots, values, digests, tags, length fields)
    for pos in rng.choice(len(proof), size=160, replace=False):
        bad = bytearray(proof)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        verdicts[_agree(o, eng, ocfg, ecfg, bytes(bad))] += 1
    # truncations and an appended tail
    for cut in (0, 1, 33, 34, 33 * 4 + 5, len(proof) // 2, len(proof) - 1):
        verdicts[_agree(o, eng, ocfg, ecfg, proof[:cut])] += 1
    verdicts[_agree(o, eng, ocfg, ecfg, proof + b"\x07")] += 1
    # an unreduced value in a triple (value + p): the arithmetic is mod p, the leaf hash is of the raw bytes
    objs_off = 33 * o.fri_num_rounds(ocfg) + 9 + 8 * (n >> (o.fri_num_rounds(ocfg) - 1))
    bad = bytearray(proof)
    v = int.from_bytes(bad[objs_off + 9:objs_off + 17], "little") + P
    bad[objs_off + 9:objs_off + 17] = v.to_bytes(8, "little")
    verdicts[_agree(o, eng, ocfg, ecfg, bytes(bad))] += 1
    assert verdicts[False] > 100 and verdicts[True] >= 1        # appending a byte with an unknown tag changes nothing


def test_fri_verify_wrong_parameters(eng, oracle):
    o = oracle
    n, exp, t, offset = 128, 4, 4, 13
    omega, ocfg, proof = _proof(o, n, exp, t, offset, 9)
    # another offset / expansion factor / test count: rejected like the reference rejects it
    for off2, exp2, t2 in ((offset + 1, exp, t), (offset, 8, t), (offset, exp, t + 1)):
        assert _agree(o, eng, o.fri_cfg(omega, off2, n, exp2, t2), eng.fri_cfg(omega, off2, n, exp2, t2), proof) is False


@pytest.mark.parametrize("which", ["ref_prime", "second_prime"])
def test_stark_verify(oracle, which):
    import stark_rs_amd as s
    o = oracle
    p, g = (P, G) if which == "ref_prime" else (P2, G2)
    e = s.Engine(p, g, 0)
    logn, lb, W, t = 10, 3, 4, 8
    n = 1 << logn
    cols = np.stack([o.splitmix64(0x5354524B00 + c, n) % np.uint64(p) for c in range(W)])
    d = e.dev_alloc(W * n * 4)
    e.dev_upload(cols.reshape(-1), d)
    # a proof without column openings is Fri::prove's bytes: nothing in it refers to the column roots, so
    # smi_stark_verify refuses it with a status of its own instead of "accepting" a proof it cannot bind
    res = e.dev_stark_prove(d, W, logn, lb, t, open_columns=False)
    roots = [bytes(r) for r in res["column_roots"]]
    with pytest.raises(s.StarkMiError) as ei:
        e.stark_verify(res["proof"], roots, W, logn, lb, t, open_columns=False)
    assert ei.value.status == -54 and "column" in str(ei.value)
    N = n << lb
    fcfg = e.fri_cfg(e.prim_nth_root(N), g, N, 1 << lb, t)
    assert e.fri_verify(fcfg, res["proof"])[0]                       # ... it still is a valid FRI proof
    res = e.dev_stark_prove(d, W, logn, lb, t, open_columns=True)
    roots = [bytes(r) for r in res["column_roots"]]
    ok, why = e.stark_verify(res["proof"], roots, W, logn, lb, t, open_columns=True)
    assert ok, why
    bad = bytearray(res["proof"])
    bad[len(bad) // 3] ^= 4
    assert not e.stark_verify(bytes(bad), roots, W, logn, lb, t, open_columns=True)[0]
    bad = bytearray(res["proof"])
    bad[-5] ^= 1                                                   # inside the last authentication path
    assert not e.stark_verify(bytes(bad), roots, W, logn, lb, t, open_columns=True)[0]
    assert not e.stark_verify(res["proof"], roots[1:] + roots[:1], W, logn, lb, t, open_columns=True)[0]
    assert not e.stark_verify(res["proof"][:-1], roots, W, logn, lb, t, open_columns=True)[0]
    # a proof made from OTHER columns is not accepted against these roots (and vice versa)
    other = np.stack([o.splitmix64(0x77 + c, n) % np.uint64(p) for c in range(W)])
    d2 = e.dev_alloc(W * n * 4)
    e.dev_upload(other.reshape(-1), d2)
    res2 = e.dev_stark_prove(d2, W, logn, lb, t, open_columns=True)
    roots2 = [bytes(r) for r in res2["column_roots"]]
    assert e.stark_verify(res2["proof"], roots2, W, logn, lb, t, open_columns=True)[0]
    assert not e.stark_verify(res2["proof"], roots, W, logn, lb, t, open_columns=True)[0]
    assert not e.stark_verify(res["proof"], roots2, W, logn, lb, t, open_columns=True)[0]
    e.dev_free(d)
    e.dev_free(d2)
    e.close()


def test_stark_prove_with_openings_does_not_need_the_callers_index_buffer(oracle):
    """smi_dev_stark_prove(open_columns = 1, top_indices = NULL): legal like in every other configuration -- the
    indices the openings need are kept internally (it used to fail with BAD_ARG after all the GPU work)."""
    import ctypes as C
    import stark_rs_amd as s
    from stark_rs_amd import _lib
    o = oracle
    e = s.Engine(P, G, 0)
    logn, lb, W, t = 8, 2, 3, 4
    n = 1 << logn
    cols = np.stack([o.splitmix64(0x99 + c, n) % np.uint64(P) for c in range(W)])
    d = e.dev_alloc(W * n * 4)
    e.dev_upload(cols.reshape(-1), d)
    want = e.dev_stark_prove(d, W, logn, lb, t, open_columns=True)
    cfg = _lib.StarkCfg(logn, lb, W, 0, 1, G, t, 1)
    proof, plen = C.c_void_p(), C.c_size_t()
    st = e.L.smi_dev_stark_prove(e.h, C.byref(cfg), C.c_void_p(d), None, C.byref(proof), C.byref(plen), None, None)
    assert st == 0
    got = C.string_at(proof, plen.value)
    e.L.smi_free(proof)
    assert got == want["proof"]
    e.dev_free(d)
    e.close()


def test_fri_verify_documented_status_not_verdict_cases(eng, oracle):
    """The two deviations include/stark_mi.h documents for malformed proofs.  (1) A last codeword of another
    length whose root matches: the reference interpolates over a truncated / repeating point list (verdict or
    "no inverse" panic); the NTT cannot, so the call returns SMI_ERR_NOT_GEOMETRIC -- a status, never accept.
    (2) An unreduced triple value far above p: FiniteField::sub's u128 `p + l - r` wraps in a release build;
    same arithmetic here, so verdict and reason equal the oracle's."""
    import stark_rs_amd as s
    o = oracle
    n, exp, t, offset = 256, 8, 5, 17
    omega, ocfg, proof = _proof(o, n, exp, t, offset, 5)
    ecfg = eng.fri_cfg(omega, offset, n, exp, t)
    R = o.fri_num_rounds(ocfg)
    n_last = n >> (R - 1)
    for new_len in (n_last // 2, n_last * 2):
        last = np.zeros(new_len, dtype=np.uint64)                  # the zero polynomial: low degree on any domain
        root = o.merkle_commit(o.leaf_hashes(last))
        forged = proof[:33 * (R - 1)] + b"\x00" + bytes(root) + b"\x02" + int(new_len).to_bytes(8, "little") + last.tobytes()
        with pytest.raises(s.StarkMiError) as ei:
            eng.fri_verify(ecfg, forged)
        assert ei.value.status == -53
    objs_off = 33 * R + 9 + 8 * n_last
    for which in (0, 1, 2):
        bad = bytearray(proof)
        bad[objs_off + 9 + 8 * which:objs_off + 17 + 8 * which] = (2 ** 64 - 1 - which).to_bytes(8, "little")
        assert _agree(o, eng, ocfg, ecfg, bytes(bad)) is False


def test_fri_verify_reports_the_first_failure_in_the_reference_order(eng, oracle):
    """two corrupted authentication paths: the reference walks (test 0: a, b, c), (test 1: a, b, c), ... and
    stops at the first that fails; the batched verifier must name the same one (src/fri.rs:431-498)."""
    o = oracle
    n, exp, t, offset = 256, 8, 5, 17
    omega, ocfg, proof = _proof(o, n, exp, t, offset, 5)
    ecfg = eng.fri_cfg(omega, offset, n, exp, t)
    R = o.fri_num_rounds(ocfg)
    depth = n.bit_length() - 1
    paths0 = 33 * R + 9 + 8 * (n >> (R - 1)) + 33 * t             # first MerklePath of layer 0
    pa, pc = 9 + 32 * depth, 9 + 32 * (depth - 1)
    at = lambda s, w: paths0 + s * (2 * pa + pc) + (0, pa, 2 * pa)[w] + 9 + 5     # a byte inside that path's first digest
    for first, second in (((0, 1), (1, 0)), ((0, 2), (1, 0)), ((1, 1), (2, 0)), ((3, 0), (3, 2))):
        bad = bytearray(proof)
        bad[at(*first)] ^= 1
        bad[at(*second)] ^= 1
        ok, _pv, why = eng.fri_verify(ecfg, bytes(bad))
        assert not ok and not o.fri_verify(ocfg, bytes(bad))
        assert why == o.fri_last_reject() == "merkle authentication path verification fails for " + ("aa", "bb", "cc")[first[1]]
