"""Golden fixtures (tests/golden/golden.json, minted by tests/golden/make_golden.py with the oracle).
CPU: the oracle still reproduces them (regression pin; the small ones also match the survey's
independently produced digests).  GPU: the HIP path matches them, including sizes whose oracle
run is too slow to repeat in every test (SHA-256 of whole outputs)."""
import hashlib
import json
import os

import numpy as np
import pytest

P, G = 998244353, 3
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def vals(o, seed, n, p=P):
    return o.splitmix64(seed, n) % np.uint64(p)


# ------------------------------------------------------------------ CPU: oracle vs fixtures
def test_oracle_reproduces_small_fixtures(oracle):
    o = oracle
    for m, d in GOLD["hash_bytes"].items():
        assert o.hash_from_bytes(bytes.fromhex(m)).hex() == d
    for e, d in GOLD["hash_field_elements"].items():
        assert o.hash_from_field_elements([int(x) for x in e.split(",")]).hex() == d
    assert GOLD["hash_bytes"]["68656c6c6f"] == "663afaa74185a1693451aa7fd22ac722ff8f89aabc0471f28dc7c2b7354cae8e"  # SURVEY 8c "hello"
    assert GOLD["merkle_root_8_from_bytes_i"] == "d86d7c3c1368c029ff23248875ffb2fb673459897e3dcbd67ac0e09ca4cdd738"
    for k, w in GOLD["roots"].items():
        assert o.ff_prim_nth_root(1 << int(k)) == w
    kat = GOLD["intt_n8_seed1_offset3"]
    assert [int(x) for x in o.fast_intt(kat["values"], o.ff_prim_nth_root(8), 3)] == kat["coeffs"]
    dom = [o.ff_mul(3, o.ff_exp(o.ff_prim_nth_root(8), k)) for k in range(8)]
    assert [int(x) for x in o.poly_interpolate_domain(dom, kat["values"])] == kat["coeffs"]   # the O(n^3) path too
    for nn in ("32", "64", "128", "256"):
        f = GOLD["fri"][nn]
        om = o.ff_prim_nth_root(int(nn))
        dm = [o.ff_mul(f["offset"], o.ff_exp(om, i)) for i in range(int(nn))]
        cfg = o.fri_cfg(om, f["offset"], int(nn), f["expansion"], f["t"])
        proof, top = o.fri_prove(cfg, o.poly_eval_domain(f["coeffs"], dm))
        assert proof.hex() == f["proof_hex"] and top == f["top_indices"]
        assert o.fri_verify(cfg, bytes.fromhex(f["proof_hex"]))


def test_oracle_reproduces_ntt_fixture_2p16(oracle):
    o = oracle
    f = GOLD["ntt"]["16"]
    n = 1 << 16
    v = vals(o, f["seed"], n)
    assert sha(o.fast_intt(v, o.ff_prim_nth_root(n), f["offset"])) == f["intt_sha256"]


# ------------------------------------------------------------------ GPU: HIP path vs fixtures
@pytest.fixture(scope="module")
def eng():
    import stark_rs_amd as s
    e = s.Engine(P, G, 0)
    yield e
    e.close()


def _trimmed(c):
    c = [int(v) for v in c]
    while c and c[-1] == 0:
        c.pop()
    return c


def test_oracle_reproduces_poly_div_fixtures(oracle):
    o = oracle
    kat = GOLD["poly_div_9_by_5"]
    assert o.poly_div(kat["a"], kat["b"]) == (kat["q"], kat["r"])
    big = GOLD["poly_div_700_by_325"]
    q, r = o.poly_div(o.splitmix64(big["seed_a"], 700) % np.uint64(P), o.splitmix64(big["seed_b"], 325) % np.uint64(P))
    assert (sha(q), sha(_trimmed(r)), len(q)) == (big["q_sha256"], big["r_sha256"], big["len_q"])


@pytest.mark.gpu
def test_gpu_poly_div_fixtures(eng, oracle):
    kat = GOLD["poly_div_9_by_5"]
    q, r = eng.poly_div(kat["a"], kat["b"])
    assert [int(v) for v in q] == kat["q"] and _trimmed(r) == _trimmed(kat["r"])
    big = GOLD["poly_div_700_by_325"]
    q, r = eng.poly_div(oracle.splitmix64(big["seed_a"], 700) % np.uint64(P), oracle.splitmix64(big["seed_b"], 325) % np.uint64(P))
    assert (sha(q), sha(_trimmed(r)), len(q), len(_trimmed(r))) == (big["q_sha256"], big["r_sha256"], big["len_q"], big["len_r"])


@pytest.mark.gpu
def test_gpu_small_fixtures(eng):
    for m, d in GOLD["hash_bytes"].items():
        assert eng.hash_bytes(bytes.fromhex(m)).hex() == d
    for e, d in GOLD["hash_field_elements"].items():
        el = [int(x) for x in e.split(",")]
        if len(el) == 1 and el[0] < P:
            assert bytes(eng.hash_leaves(el)[0]).hex() == d
    kat = GOLD["intt_n8_seed1_offset3"]
    assert [int(x) for x in eng.intt(kat["values"], 3)] == kat["coeffs"]
    for k, w in GOLD["roots"].items():
        assert eng.prim_nth_root(1 << int(k)) == w


@pytest.mark.gpu
@pytest.mark.parametrize("logn", ["10", "16", "20", "23"])
def test_gpu_ntt_fixtures(eng, oracle, logn):
    f = GOLD["ntt"][logn]
    n = 1 << int(logn)
    v = vals(oracle, f["seed"], n)
    assert sha(eng.intt(v, f["offset"])) == f["intt_sha256"]
    assert sha(eng.coset_ntt(v[: n // 8], int(logn), f["offset"])) == f["coset_ntt_of_first_eighth_sha256"]


@pytest.mark.gpu
def test_gpu_second_prime_fixture(oracle):
    import stark_rs_amd as s
    f = GOLD["ntt_p2_24"]
    e2 = s.Engine(f["prime"], 3, 0)
    v = vals(oracle, f["seed"], 1 << 24, f["prime"])
    assert sha(e2.intt(v, f["offset"])) == f["intt_sha256"]
    e2.close()


@pytest.mark.gpu
def test_gpu_lde_cfg3_fixture(eng, oracle):
    """BASELINE configs[2] at full size: 2^20 rows x 4 columns -> 2^23, every output element (SHA-256)."""
    n = 1 << 20
    cols = np.stack([vals(oracle, 0x5354524B00 + c, n) for c in range(4)])
    out = eng.lde(cols, 3, 1, G)
    for c in range(4):
        assert sha(out[c]) == GOLD["lde_cfg3_sha256"][str(c)]


@pytest.mark.gpu
def test_gpu_fri_fixtures(eng, oracle):
    o = oracle
    for nn in ("32", "64", "128", "256"):
        f = GOLD["fri"][nn]
        om = o.ff_prim_nth_root(int(nn))
        dm = [o.ff_mul(f["offset"], o.ff_exp(om, i)) for i in range(int(nn))]
        cfg = eng.fri_cfg(om, f["offset"], int(nn), f["expansion"], f["t"])
        cw = o.poly_eval_domain(f["coeffs"], dm)
        proof, top = eng.fri_prove(cfg, cw)
        assert proof.hex() == f["proof_hex"] and top == f["top_indices"]
        roots, alphas, _ = eng.fri_commit(cfg, cw)
        assert [bytes(r).hex() for r in roots] == f["roots"] and alphas == f["alphas"]
    f = GOLD["fri"]["65536"]
    nn = 1 << 16
    om = o.ff_prim_nth_root(nn)
    cw = eng.coset_ntt(vals(o, f["coeff_seed"], nn // f["expansion"]), 16, f["offset"])
    proof, top = eng.fri_prove(eng.fri_cfg(om, f["offset"], nn, f["expansion"], f["t"]), cw)
    assert len(proof) == f["proof_len"] and hashlib.sha256(proof).hexdigest() == f["proof_sha256"] and top == f["top_indices"]
