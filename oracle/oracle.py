"""ctypes binding of the CPU oracle (oracle/stark_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package (stark_rs_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SMI_ORACLE_LIB") or os.path.join(_HERE, "build", "libstark_oracle.so")   # override: the sanitizer build (`make -C oracle asan`)

P_REF = 998244353          # src/ff.rs:192
P2 = 469762049             # 7*2^26+1, generator 3 (not a reference constant)
G_REF, G2 = 3, 3


class OraclePanic(Exception):
    """The reference would have panicked with this message."""


def build(force=False):
    if os.environ.get("SMI_ORACLE_LIB"):
        return _SO
    src = [os.path.join(_HERE, f) for f in ("stark_oracle.c", "stark_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class FriCfg(C.Structure):
    _fields_ = [("p", u64), ("omega", u64), ("offset", u64), ("domain_length", u64),
                ("expansion_factor", u64), ("num_colinearity_tests", u64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.so_last_panic.restype = C.c_char_p
        L.so_fri_last_reject.restype = C.c_char_p
        for name in ("so_ff_add", "so_ff_sub", "so_ff_mul", "so_ff_div", "so_ff_exp", "so_ff_prim_nth_root"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64, u64, u64]
        L.so_ff_prim_nth_root.argtypes = [u64, u64]
        for name in ("so_ff_neg", "so_ff_inv"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64, u64]
        L.so_ff_g.restype = u64
        L.so_ff_g.argtypes = [u64]
        L.so_ff_prim_nth_root_g.restype = u64
        L.so_ff_prim_nth_root_g.argtypes = [u64, u64, u64]
        L.so_ff_sample.restype = u64
        L.so_ff_sample.argtypes = [u64, C.c_char_p, C.c_size_t]
        L.so_xgcd.argtypes = [u64, u64, C.POINTER(C.c_int64)]
        L.so_poly_deg.restype = C.c_int64
        L.so_poly_deg.argtypes = [u64p, C.c_size_t]
        L.so_poly_eq.argtypes = [u64, u64p, C.c_size_t, u64p, C.c_size_t]
        for name in ("so_poly_add", "so_poly_sub", "so_poly_mul"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [u64, u64p, C.c_size_t, u64p, C.c_size_t, u64p]
        L.so_poly_scale.restype = C.c_size_t
        L.so_poly_scale.argtypes = [u64, u64p, C.c_size_t, u64, u64p]
        L.so_poly_eval.restype = u64
        L.so_poly_eval.argtypes = [u64, u64p, C.c_size_t, u64]
        L.so_poly_eval_domain.argtypes = [u64, u64p, C.c_size_t, u64p, C.c_size_t, u64p]
        L.so_poly_interpolate_domain.restype = C.c_size_t
        L.so_poly_interpolate_domain.argtypes = [u64, u64p, u64p, C.c_size_t, u64p]
        L.so_poly_zerofier.restype = C.c_size_t
        L.so_poly_zerofier.argtypes = [u64, u64p, C.c_size_t, u64p]
        L.so_poly_div.argtypes = [u64, u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.POINTER(C.c_size_t), u64p,
                                  C.POINTER(C.c_size_t)]
        L.so_poly_exp.restype = C.c_size_t
        L.so_poly_exp.argtypes = [u64, u64p, C.c_size_t, u64, u64p, C.c_size_t]
        L.so_poly_test_colinearity.argtypes = [u64, u64p, u64p, C.c_size_t]
        L.so_hash_from_bytes.argtypes = [C.c_char_p, C.c_size_t, u8p]
        L.so_hash_from_field_elements.argtypes = [u64p, C.c_size_t, u8p]
        L.so_hash_from_u64.argtypes = [u64, u8p]
        L.so_hash_combine.argtypes = [C.c_char_p, C.c_char_p, u8p]
        L.so_merkle_new.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.so_merkle_commit.argtypes = [C.c_void_p, C.c_size_t, u8p]
        L.so_merkle_open.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.so_merkle_verify.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
        L.so_fs_new.restype = C.c_void_p
        L.so_fs_free.argtypes = [C.c_void_p]
        L.so_fs_absorb.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.so_fs_challenge.restype = u64
        L.so_fs_challenge.argtypes = [C.c_void_p]
        L.so_trace_to_field_elements.argtypes = [u64p, u64p, C.c_size_t, u64p]
        L.so_trace_fibonacci.argtypes = [C.c_size_t, u64p, u64p]
        L.so_fri_new_check.argtypes = [C.POINTER(FriCfg)]
        L.so_fri_num_rounds.restype = u64
        L.so_fri_num_rounds.argtypes = [C.POINTER(FriCfg)]
        L.so_fri_fold_codeword.argtypes = [C.POINTER(FriCfg), u64p, C.c_size_t, u64, u64, u64, u64p]
        L.so_fri_eval_domain.argtypes = [C.POINTER(FriCfg), C.c_size_t, u64p]
        L.so_fri_sample_index.restype = C.c_size_t
        L.so_fri_sample_index.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t]
        L.so_fri_sample_indices.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, u64p]
        L.so_fri_prove.argtypes = [C.POINTER(FriCfg), u64p, C.c_size_t, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), u64p]
        L.so_fri_commit_trace.argtypes = [C.POINTER(FriCfg), u64p, C.c_size_t, C.c_void_p, u64p, u64p,
                                          C.POINTER(C.c_size_t)]
        L.so_fri_verify.argtypes = [C.POINTER(FriCfg), C.c_char_p, C.c_size_t, u64p, u64p, C.POINTER(C.c_size_t)]
        L.so_free.argtypes = [C.c_void_p]
        L.so_fast_intt.argtypes = [u64, u64, u64, u64p, C.c_size_t, u64p]
        L.so_fast_coset_ntt.argtypes = [u64, u64, u64, u64p, C.c_size_t, C.c_size_t, u64p]
        L.so_fast_fold.argtypes = [u64, u64p, C.c_size_t, u64, u64, u64, u64p]
        _lib = L
    return _lib


def _chk():
    L = lib()
    msg = L.so_last_panic()
    if msg:
        L.so_clear_panic()
        raise OraclePanic(msg.decode())


def _arr(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    return a


def _p(a):
    return a.ctypes.data_as(u64p)


# ------------------------------------------------------------------ field
def ff_add(l, r, p=P_REF): return int(lib().so_ff_add(p, l, r))
def ff_sub(l, r, p=P_REF): return int(lib().so_ff_sub(p, l, r))
def ff_mul(l, r, p=P_REF): return int(lib().so_ff_mul(p, l, r))
def ff_neg(x, p=P_REF): return int(lib().so_ff_neg(p, x))


def ff_inv(x, p=P_REF):
    v = int(lib().so_ff_inv(p, x)); _chk(); return v


def ff_div(l, r, p=P_REF):
    v = int(lib().so_ff_div(p, l, r)); _chk(); return v


def ff_exp(b, e, p=P_REF): return int(lib().so_ff_exp(p, b, e))


def ff_g(p=P_REF):
    v = int(lib().so_ff_g(p)); _chk(); return v


def ff_prim_nth_root(n, p=P_REF):
    v = int(lib().so_ff_prim_nth_root(p, n)); _chk(); return v


def ff_prim_nth_root_g(n, p, g):
    v = int(lib().so_ff_prim_nth_root_g(p, g, n)); _chk(); return v


def ff_sample(salt: bytes, p=P_REF): return int(lib().so_ff_sample(p, salt, len(salt)))


def xgcd(x, y):
    out = (C.c_int64 * 3)()
    lib().so_xgcd(x, y, out)
    return tuple(int(v) for v in out)


# ------------------------------------------------------------- polynomials
def poly_deg(c):
    a = _arr(c); return int(lib().so_poly_deg(_p(a), len(a)))


def poly_eq(a, b, p=P_REF):
    a, b = _arr(a), _arr(b); return bool(lib().so_poly_eq(p, _p(a), len(a), _p(b), len(b)))


def _binop(fn, a, b, p, cap):
    a, b = _arr(a), _arr(b)
    out = np.zeros(max(cap, 1), dtype=np.uint64)
    n = fn(p, _p(a), len(a), _p(b), len(b), _p(out))
    _chk()
    return [int(v) for v in out[:n]]


def poly_add(a, b, p=P_REF): return _binop(lib().so_poly_add, a, b, p, max(len(a), len(b)))
def poly_sub(a, b, p=P_REF): return _binop(lib().so_poly_sub, a, b, p, max(len(a), len(b)))
def poly_mul(a, b, p=P_REF): return _binop(lib().so_poly_mul, a, b, p, len(a) + len(b))


def poly_scale(a, factor, p=P_REF):
    a = _arr(a); out = np.zeros(max(len(a), 1), dtype=np.uint64)
    n = lib().so_poly_scale(p, _p(a), len(a), factor, _p(out))
    return [int(v) for v in out[:n]]


def poly_eval(c, x, p=P_REF):
    a = _arr(c); return int(lib().so_poly_eval(p, _p(a), len(a), x))


def poly_eval_domain(c, dom, p=P_REF):
    a, d = _arr(c), _arr(dom)
    out = np.zeros(len(d), dtype=np.uint64)
    lib().so_poly_eval_domain(p, _p(a), len(a), _p(d), len(d), _p(out))
    return out


def poly_interpolate_domain(dom, vals, p=P_REF):
    d, v = _arr(dom), _arr(vals)
    if len(d) != len(v):
        raise OraclePanic("assertion failed: domain.len() == values.len()")
    out = np.zeros(max(len(d), 1), dtype=np.uint64)
    n = lib().so_poly_interpolate_domain(p, _p(d), _p(v), len(d), _p(out))
    _chk()
    return out[:n].copy()


def poly_zerofier(dom, p=P_REF):
    d = _arr(dom); out = np.zeros(len(d) + 2, dtype=np.uint64)
    n = lib().so_poly_zerofier(p, _p(d), len(d), _p(out))
    return [int(v) for v in out[:n]]


def poly_div(a, b, p=P_REF):
    a, b = _arr(a), _arr(b)
    q = np.zeros(len(a) + 2, dtype=np.uint64); r = np.zeros(len(a) + len(b) + 2, dtype=np.uint64)
    nq, nr = C.c_size_t(), C.c_size_t()
    lib().so_poly_div(p, _p(a), len(a), _p(b), len(b), _p(q), C.byref(nq), _p(r), C.byref(nr))
    _chk()
    return [int(v) for v in q[:nq.value]], [int(v) for v in r[:nr.value]]


def poly_exp(a, e, p=P_REF):
    a = _arr(a); cap = max(1, (len(a) - 1) * max(e, 1) * 2 + 2)
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().so_poly_exp(p, _p(a), len(a), e, _p(out), cap)
    _chk()
    return [int(v) for v in out[:n]]


def poly_test_colinearity(points, p=P_REF):
    xs = _arr([x for x, _ in points]); ys = _arr([y for _, y in points])
    r = bool(lib().so_poly_test_colinearity(p, _p(xs), _p(ys), len(xs))); _chk(); return r


# -------------------------------------------------------------------- hash
def hash_from_bytes(b: bytes) -> bytes:
    out = (C.c_uint8 * 32)(); lib().so_hash_from_bytes(b, len(b), out); return bytes(out)


def hash_from_field_elements(e) -> bytes:
    a = _arr(e); out = (C.c_uint8 * 32)(); lib().so_hash_from_field_elements(_p(a), len(a), out); return bytes(out)


def hash_from_u64(v) -> bytes:
    out = (C.c_uint8 * 32)(); lib().so_hash_from_u64(v, out); return bytes(out)


def hash_combine(l: bytes, r: bytes) -> bytes:
    out = (C.c_uint8 * 32)(); lib().so_hash_combine(l, r, out); return bytes(out)


def leaf_hashes(codeword) -> np.ndarray:
    """fri.rs:118-121: one element per leaf -> (n, 32) uint8."""
    a = _arr(codeword)
    out = np.zeros((len(a), 32), dtype=np.uint8)
    tmp = (C.c_uint8 * 32)()
    L = lib()
    for i in range(len(a)):
        L.so_hash_from_field_elements(_p(a[i:i + 1]), 1, tmp)
        out[i] = np.frombuffer(bytes(tmp), dtype=np.uint8)
    return out


# ------------------------------------------------------------------ merkle
def merkle_new(leaves: np.ndarray) -> np.ndarray:
    """leaves (n,32) uint8 -> nodes (2n-1, 32), levels back to back."""
    leaves = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
    n = len(leaves)
    nodes = np.zeros((max(2 * n - 1, 1), 32), dtype=np.uint8)
    lib().so_merkle_new(leaves.ctypes.data, n, nodes.ctypes.data)
    _chk()
    return nodes


def merkle_commit(leaves) -> bytes:
    leaves = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
    out = (C.c_uint8 * 32)()
    lib().so_merkle_commit(leaves.ctypes.data, len(leaves), out)
    _chk()
    return bytes(out)


def merkle_open(nodes: np.ndarray, n: int, index: int):
    path = np.zeros((64, 32), dtype=np.uint8)
    d = lib().so_merkle_open(nodes.ctypes.data, n, index, path.ctypes.data)
    _chk()
    return [bytes(path[i]) for i in range(d)]


def merkle_verify(leaf: bytes, index: int, path, root: bytes) -> bool:
    return bool(lib().so_merkle_verify(leaf, index, b"".join(path), len(path), root))


# ------------------------------------------------------------- fiat-shamir
class FiatShamir:
    def __init__(self):
        self._h = lib().so_fs_new()

    def absorb(self, data: bytes):
        lib().so_fs_absorb(self._h, data, len(data))

    def challenge(self) -> int:
        return int(lib().so_fs_challenge(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().so_fs_free(self._h); self._h = None


# --------------------------------------------------------------------- FRI
def fri_cfg(omega, offset, domain_length, expansion_factor, num_colinearity_tests, p=P_REF):
    c = FriCfg(p, omega, offset, domain_length, expansion_factor, num_colinearity_tests)
    lib().so_fri_new_check(C.byref(c)); _chk()
    return c


def fri_num_rounds(cfg): return int(lib().so_fri_num_rounds(C.byref(cfg)))


def fri_fold_codeword(cfg, codeword, alpha, offset, omega):
    a = _arr(codeword); out = np.zeros(len(a) // 2, dtype=np.uint64)
    lib().so_fri_fold_codeword(C.byref(cfg), _p(a), len(a), alpha, offset, omega, _p(out))
    _chk()
    return out


def fri_eval_domain(cfg, rnd=0):
    out = np.zeros(cfg.domain_length >> rnd, dtype=np.uint64)
    lib().so_fri_eval_domain(C.byref(cfg), rnd, _p(out)); return out


def fri_sample_index(b: bytes, size): return int(lib().so_fri_sample_index(b, len(b), size))


def fri_sample_indices(seed: bytes, size, reduced_size, number):
    out = np.zeros(max(number, 1), dtype=np.uint64)
    lib().so_fri_sample_indices(seed, len(seed), size, reduced_size, number, _p(out)); _chk()
    return [int(v) for v in out[:number]]


def fri_prove(cfg, codeword):
    """-> (serialized proof bytes, top-level indices)."""
    a = _arr(codeword)
    proof = C.c_void_p(); plen = C.c_size_t()
    idx = np.zeros(max(cfg.num_colinearity_tests, 1), dtype=np.uint64)
    rc = lib().so_fri_prove(C.byref(cfg), _p(a), len(a), C.byref(proof), C.byref(plen), _p(idx))
    _chk()
    assert rc == 0
    b = C.string_at(proof, plen.value)
    lib().so_free(proof)
    return b, [int(v) for v in idx[:cfg.num_colinearity_tests]]


def fri_commit_trace(cfg, codeword):
    """-> (roots (R,32) uint8, alphas list[R-1], last codeword)."""
    a = _arr(codeword)
    R = fri_num_rounds(cfg)
    roots = np.zeros((max(R, 1), 32), dtype=np.uint8)
    alphas = np.zeros(max(R, 1), dtype=np.uint64)
    last = np.zeros(len(a), dtype=np.uint64)
    ll = C.c_size_t()
    lib().so_fri_commit_trace(C.byref(cfg), _p(a), len(a), roots.ctypes.data, _p(alphas), _p(last), C.byref(ll))
    _chk()
    return roots[:R], [int(v) for v in alphas[:max(R - 1, 0)]], last[:ll.value].copy()


def fri_verify(cfg, proof: bytes, want_values=False):
    t = cfg.num_colinearity_tests
    pi = np.zeros(2 * t + 2, dtype=np.uint64); pv = np.zeros(2 * t + 2, dtype=np.uint64); n = C.c_size_t()
    r = lib().so_fri_verify(C.byref(cfg), proof, len(proof), _p(pi), _p(pv), C.byref(n))
    _chk()
    if want_values:
        return bool(r), [(int(pi[i]), int(pv[i])) for i in range(n.value)]
    return bool(r)


def fri_last_reject(): return lib().so_fri_last_reject().decode()


# ------------------------------------------------- fast CPU NTT restatement
def fast_intt(vals, omega, offset=1, p=P_REF):
    a = _arr(vals); out = np.zeros(len(a), dtype=np.uint64)
    lib().so_fast_intt(p, omega, offset, _p(a), len(a), _p(out)); return out


def fast_coset_ntt(coeffs, N, omega_N, offset=1, p=P_REF):
    a = _arr(coeffs); out = np.zeros(N, dtype=np.uint64)
    lib().so_fast_coset_ntt(p, omega_N, offset, _p(a), len(a), N, _p(out)); return out


def fast_fold(codeword, alpha, offset, omega, p=P_REF):
    a = _arr(codeword); out = np.zeros(len(a) // 2, dtype=np.uint64)
    lib().so_fast_fold(p, _p(a), len(a), alpha, offset, omega, _p(out)); return out


# --------------------------------------------------------- synthetic inputs
def splitmix64(seed: int, n: int) -> np.ndarray:
    """value(seed,i) = splitmix64(seed + i); returns raw u64 (reduce with % p)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z
