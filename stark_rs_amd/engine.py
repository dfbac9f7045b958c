"""Engine: one smi_ctx (one GPU, one prime field) with numpy- and pointer-level calls.

Host-buffer methods take/return numpy uint64 arrays (the reference's wire width); dev_*
methods take raw device pointers (ints, e.g. torch.Tensor.data_ptr()) of u32 residues and
only enqueue work on the context's stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FriCfg, StarkMiError, check, vp

P_REF, G_REF = 998244353, 3          # reference field (src/ff.rs:191-197)
P2, G2 = 469762049, 3              # 7*2^26+1: domains above 2^23; p < 2^30 keeps 4p < 2^32

_default = {}


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


class Engine:
    def __init__(self, p=P_REF, g=G_REF, device=0):
        self.L = _lib.lib()
        h = vp()
        st = self.L.smi_ctx_create(p, g, device, C.byref(h))
        if st != 0:
            raise StarkMiError(st, f"smi_ctx_create(p={p}, g={g}, device={device}): {_lib.status_string(st)} "
                                   "(the HIP path is the only path: no CPU fallback)")
        self.h = h
        self.p, self.g, self.device = p, g, device

    def close(self):
        if getattr(self, "h", None):
            self.L.smi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        check(st, self.h)

    # ---- context
    def set_stream(self, stream_ptr):
        self._ck(self.L.smi_ctx_set_stream(self.h, vp(stream_ptr)))

    def sync(self):
        self._ck(self.L.smi_ctx_sync(self.h))

    def profile(self, enable=True):
        """Bracket every hot-path kernel launch with HIP events on the context's stream."""
        self._ck(self.L.smi_ctx_profile(self.h, 1 if enable else 0))

    def profile_only(self, name_part=None):
        """Bracket only launches whose kernel name contains name_part (None: all): one kernel timed inside an otherwise
        undisturbed loop."""
        self._ck(self.L.smi_ctx_profile_only(self.h, name_part.encode() if name_part else None))

    def lde_two_pass(self, enable=True):
        """extensions of 2^20..2^22 rows in two passes over the outputs (csrc/lde_core.h) instead of three"""
        self._ck(self.L.smi_ctx_lde_two_pass(self.h, 1 if enable else 0))

    def copy_probe(self, enable=True):
        """Measurement aid: NTT passes launch their copy-only twins (same access pattern, no
        arithmetic); results are meaningless while it is on."""
        self._ck(self.L.smi_ctx_copy_probe(self.h, 1 if enable else 0))

    def mix_probe(self, mixes=512):
        """mix_state evaluations per second of the bare permutation (two hashes per lane, no memory traffic): the
        integer-VALU ceiling the Merkle kernels are reported against."""
        out = C.c_double()
        self._ck(self.L.smi_ctx_mix_probe(self.h, mixes, C.byref(out)))
        return float(out.value)

    def profile_read(self):
        """-> {kernel name: {"launches", "total_ms", "alg_bytes", "alg_mixes"}} since the last read (synchronises)."""
        arr = (_lib.KernelTime * 64)()
        n = C.c_size_t()
        self._ck(self.L.smi_ctx_profile_read(self.h, arr, 64, C.byref(n)))
        return {arr[i].name.decode(): {"launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms),
                                       "alg_bytes": float(arr[i].alg_bytes), "alg_mixes": float(arr[i].alg_mixes)}
                for i in range(n.value)}

    @property
    def two_adicity(self):
        return int(self.L.smi_ctx_two_adicity(self.h))

    # ---- scalars
    def prim_nth_root(self, n):
        out = C.c_uint64()
        self._ck(self.L.smi_prim_nth_root(self.h, n, C.byref(out)))
        return out.value

    def inv(self, x):
        out = C.c_uint64()
        self._ck(self.L.smi_ff_inv(self.h, x, C.byref(out)))
        return out.value

    def exp(self, b, e):
        out = C.c_uint64()
        self._ck(self.L.smi_ff_exp(self.h, b, e, C.byref(out)))
        return out.value

    def mul(self, a, b):
        out = C.c_uint64()
        self._ck(self.L.smi_ff_mul(self.h, a, b, C.byref(out)))
        return out.value

    # ---- univariate (host buffers)
    def intt(self, values, offset=1):
        v = _u64(values)
        n = len(v)
        if n == 0 or n & (n - 1):
            raise StarkMiError(-3, "n must be a power of two")
        out = np.empty(n, dtype=np.uint64)
        self._ck(self.L.smi_intt(self.h, v.ctypes.data, out.ctypes.data, n.bit_length() - 1, offset))
        return out

    def coset_ntt(self, coeffs, log_N, offset=1):
        c = _u64(coeffs)
        out = np.empty(1 << log_N, dtype=np.uint64)
        self._ck(self.L.smi_coset_ntt(self.h, c.ctypes.data, len(c), out.ctypes.data, log_N, offset))
        return out

    def poly_scale(self, coeffs, factor):
        c = _u64(coeffs)
        out = np.empty(len(c), dtype=np.uint64)
        self._ck(self.L.smi_poly_scale(self.h, c.ctypes.data, len(c), factor, out.ctypes.data))
        return out

    def poly_mul(self, a, b):
        """Polynomial::mul (mul.rs:6-29) by NTT."""
        a, b = _u64(a), _u64(b)
        out = np.empty(max(len(a) + len(b), 1), dtype=np.uint64)
        n = C.c_size_t()
        self._ck(self.L.smi_poly_mul(self.h, a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data, C.byref(n)))
        return out[:n.value].copy()

    def poly_div(self, a, b):
        """Polynomial::div (div.rs:6-42) -> (quotient, remainder) by NTT products (power-series inverse)."""
        a, b = _u64(a), _u64(b)
        q = np.empty(max(len(a), 1), dtype=np.uint64)
        r = np.empty(max(len(a), len(b), 1), dtype=np.uint64)
        nq, nr = C.c_size_t(), C.c_size_t()
        self._ck(self.L.smi_poly_div(self.h, a.ctypes.data, len(a), b.ctypes.data, len(b), q.ctypes.data, C.byref(nq),
                                     r.ctypes.data, C.byref(nr)))
        return q[:nq.value].copy(), r[:nr.value].copy()

    def domain_is_geometric(self, domain):
        d = _u64(domain)
        off = C.c_uint64()
        st = self.L.smi_domain_is_geometric(self.h, d.ctypes.data, len(d), C.byref(off))
        return (st == 0), off.value

    def lde(self, cols, log_blowup, trace_offset=1, lde_offset=None, out=None):
        cols = _u64(cols)
        n_cols, n = cols.shape
        if out is None:
            out = np.empty((n_cols, n << log_blowup), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.shape == (n_cols, n << log_blowup) and out.flags.c_contiguous
        lde_offset = self.g if lde_offset is None else lde_offset
        self._ck(self.L.smi_lde(self.h, cols.ctypes.data, n_cols, n.bit_length() - 1, log_blowup, trace_offset, lde_offset,
                                out.ctypes.data))
        return out

    def trace_pack(self, rows_i128_bytes, n_rows, n_cols):
        out = np.empty((n_cols, n_rows), dtype=np.uint64)
        buf = np.frombuffer(rows_i128_bytes, dtype=np.uint8)
        self._ck(self.L.smi_trace_pack(self.h, buf.ctypes.data, n_rows, n_cols, out.ctypes.data))
        return out

    # ---- hash / merkle (host buffers)
    def hash_leaves(self, elems):
        e = _u64(elems)
        out = np.empty((len(e), 32), dtype=np.uint8)
        self._ck(self.L.smi_hash_leaves(self.h, e.ctypes.data, len(e), out.ctypes.data))
        return out

    def hash_combine_pairs(self, digests):
        d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 64)
        out = np.empty((len(d), 32), dtype=np.uint8)
        self._ck(self.L.smi_hash_combine_pairs(self.h, d.ctypes.data, len(d), out.ctypes.data))
        return out

    def hash_bytes(self, msg: bytes) -> bytes:
        out = (C.c_uint8 * 32)()
        self._ck(self.L.smi_hash_bytes(self.h, msg, len(msg), out))
        return bytes(out)

    def hash_bytes_batch(self, msgs):
        """Hash::from_bytes of equally long messages -> list of 32-byte digests (one device call)."""
        msgs = list(msgs)
        if not msgs:
            return []
        ln = len(msgs[0])
        assert all(len(m) == ln for m in msgs)
        out = np.empty((len(msgs), 32), dtype=np.uint8)
        self._ck(self.L.smi_hash_bytes_batch(self.h, b"".join(msgs), len(msgs), ln, out.ctypes.data))
        return [bytes(r) for r in out]

    def merkle_commit(self, leaves) -> bytes:
        l = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
        out = (C.c_uint8 * 32)()
        self._ck(self.L.smi_merkle_commit(self.h, l.ctypes.data, len(l), out))
        return bytes(out)

    def merkle_verify_batch(self, leaves, indices, paths, root):
        """MerkleTree::verify for k triples sharing one depth and root -> bool array."""
        l = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
        k = len(l)
        pth = np.ascontiguousarray(paths, dtype=np.uint8).reshape(k, -1, 32) if k else np.zeros((0, 0, 32), np.uint8)
        idx = _u64(indices)
        ok = np.zeros(max(k, 1), dtype=np.uint8)
        rt = np.frombuffer(bytes(root), dtype=np.uint8).copy()
        self._ck(self.L.smi_merkle_verify_batch(self.h, l.ctypes.data, idx.ctypes.data, pth.ctypes.data, k, pth.shape[1],
                                                rt.ctypes.data, ok.ctypes.data))
        return ok[:k].astype(bool)

    def merkle_new(self, leaves):
        l = np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 32)
        t = vp()
        self._ck(self.L.smi_merkle_new(self.h, l.ctypes.data, len(l), C.byref(t)))
        return DeviceTree(self, t)

    def merkle_from_codeword(self, codeword):
        c = _u64(codeword)
        t = vp()
        self._ck(self.L.smi_merkle_from_codeword(self.h, c.ctypes.data, len(c), C.byref(t)))
        return DeviceTree(self, t)

    # ---- fri (host buffers)
    def fri_cfg(self, omega, offset, domain_length, expansion_factor, num_colinearity_tests):
        cfg = FriCfg(omega, offset, domain_length, expansion_factor, num_colinearity_tests)
        self._ck(self.L.smi_fri_check(self.h, C.byref(cfg)))
        return cfg

    def fri_num_rounds(self, cfg):
        r = C.c_uint64()
        self._ck(self.L.smi_fri_num_rounds(C.byref(cfg), C.byref(r)))
        return r.value

    def fri_fold(self, codeword, alpha, offset, omega):
        c = _u64(codeword)
        out = np.empty(len(c) // 2, dtype=np.uint64)
        self._ck(self.L.smi_fri_fold(self.h, c.ctypes.data, len(c), alpha, offset, omega, out.ctypes.data))
        return out

    def fri_commit(self, cfg, codeword):
        c = _u64(codeword)
        R = max(self.fri_num_rounds(cfg), 1)
        roots = np.zeros((R, 32), dtype=np.uint8)
        alphas = np.zeros(R, dtype=np.uint64)
        last = np.zeros(len(c), dtype=np.uint64)
        ll = C.c_size_t()
        self._ck(self.L.smi_fri_commit(self.h, C.byref(cfg), c.ctypes.data, len(c), roots.ctypes.data, alphas.ctypes.data,
                                       last.ctypes.data, C.byref(ll), None))
        return roots, [int(a) for a in alphas[:R - 1]], last[:ll.value].copy()

    def fri_commit_run(self, cfg, codeword):
        """Fri::commit keeping every round's codeword and tree on the device -> (roots, alphas, FriRun)."""
        c = _u64(codeword)
        R = max(self.fri_num_rounds(cfg), 1)
        roots = np.zeros((R, 32), dtype=np.uint8)
        alphas = np.zeros(R, dtype=np.uint64)
        last = np.zeros(len(c), dtype=np.uint64)
        ll, run = C.c_size_t(), vp()
        self._ck(self.L.smi_fri_commit(self.h, C.byref(cfg), c.ctypes.data, len(c), roots.ctypes.data, alphas.ctypes.data,
                                       last.ctypes.data, C.byref(ll), C.byref(run)))
        return roots, [int(a) for a in alphas[:R - 1]], FriRun(self, run)

    def fri_prove(self, cfg, codeword):
        """-> (ProofStream::serialize bytes, top-level indices) -- Fri::prove, src/fri.rs:250-311."""
        c = _u64(codeword)
        proof, plen = vp(), C.c_size_t()
        top = np.zeros(max(cfg.num_colinearity_tests, 1), dtype=np.uint64)
        self._ck(self.L.smi_fri_prove(self.h, C.byref(cfg), c.ctypes.data, len(c), C.byref(proof), C.byref(plen),
                                      top.ctypes.data))
        b = C.string_at(proof, plen.value)
        self.L.smi_free(proof)
        return b, [int(v) for v in top[:cfg.num_colinearity_tests]]

    # ---- device-resident calls (pointers are ints)
    def fri_verify(self, cfg, proof: bytes):
        """Fri::verify (src/fri.rs:313-504) -> (accept, [(index, value)], reason); a reference panic raises."""
        t = int(cfg.num_colinearity_tests)
        pi, pv = np.zeros(2 * t + 2, dtype=np.uint64), np.zeros(2 * t + 2, dtype=np.uint64)
        acc, n = C.c_int(), C.c_size_t()
        self._ck(self.L.smi_fri_verify(self.h, C.byref(cfg), proof, len(proof), C.byref(acc), pi.ctypes.data, pv.ctypes.data, C.byref(n)))
        why = "" if acc.value else self.L.smi_last_error(self.h).decode()
        return bool(acc.value), [(int(pi[i]), int(pv[i])) for i in range(n.value)], why

    def stark_verify(self, proof: bytes, column_roots, n_cols, log_n, log_blowup, num_colinearity_tests, trace_offset=1,
                     lde_offset=None, open_columns=False):
        """verifier of dev_stark_prove / MultiGpu.stark_prove -> (accept, reason)"""
        cfg = _lib.StarkCfg(log_n, log_blowup, n_cols, 0, trace_offset, self.g if lde_offset is None else lde_offset,
                            num_colinearity_tests, 1 if open_columns else 0)
        roots = np.ascontiguousarray(np.frombuffer(b"".join(bytes(r) for r in column_roots), dtype=np.uint8))
        acc = C.c_int()
        self._ck(self.L.smi_stark_verify(self.h, C.byref(cfg), roots.ctypes.data, proof, len(proof), C.byref(acc)))
        return bool(acc.value), ("" if acc.value else self.L.smi_last_error(self.h).decode())

    def dev_alloc(self, nbytes):
        d = vp()
        self._ck(self.L.smi_dev_alloc(self.h, nbytes, C.byref(d)))
        return d.value

    def dev_free(self, ptr):
        self._ck(self.L.smi_dev_free(self.h, vp(ptr)))

    def dev_upload(self, values, d_ptr, reduce=False):
        v = _u64(values)
        self._ck(self.L.smi_dev_upload_u64(self.h, v.ctypes.data, v.size, vp(d_ptr), 1 if reduce else 0))

    def dev_download(self, d_ptr, n):
        out = np.empty(n, dtype=np.uint64)
        self._ck(self.L.smi_dev_download_u64(self.h, vp(d_ptr), n, out.ctypes.data))
        return out

    def dev_ntt(self, d_in, d_out, log_n, n_in=None, batch=1, in_stride=None, out_stride=None, inverse=False, offset=1,
                post_scale=1):
        n = 1 << log_n
        n_in = n if n_in is None else n_in
        self._ck(self.L.smi_dev_ntt(self.h, vp(d_in), vp(d_out), log_n, n_in, batch, n_in if in_stride is None else in_stride,
                                    n if out_stride is None else out_stride, 1 if inverse else 0, offset, post_scale))

    def dev_lde(self, d_cols, n_cols, log_n, log_blowup, d_out, trace_offset=1, lde_offset=None):
        lde_offset = self.g if lde_offset is None else lde_offset
        self._ck(self.L.smi_dev_lde(self.h, vp(d_cols), n_cols, log_n, log_blowup, trace_offset, lde_offset, vp(d_out)))

    def dev_hash_leaves(self, d_elems, n, d_digests):
        self._ck(self.L.smi_dev_hash_leaves(self.h, vp(d_elems), n, vp(d_digests)))

    def dev_merkle_build(self, d_elems, n, d_nodes):
        self._ck(self.L.smi_dev_merkle_build(self.h, vp(d_elems), n, vp(d_nodes)))

    def dev_merkle_build_rows(self, d_cols, n_cols, col_stride, n, d_nodes):
        """One tree whose leaf i is Hash::from_field_elements(row i) over n_cols columns."""
        self._ck(self.L.smi_dev_merkle_build_rows(self.h, vp(d_cols), n_cols, col_stride, n, vp(d_nodes)))

    def dev_hash_bytes(self, d_msg, length, d_out32):
        self._ck(self.L.smi_dev_hash_bytes(self.h, vp(d_msg), length, vp(d_out32)))

    def dev_merkle_from_digests(self, n, d_nodes):
        self._ck(self.L.smi_dev_merkle_from_digests(self.h, n, vp(d_nodes)))

    def dev_fri_fold(self, d_in, length, d_alpha, offset, omega, d_out):
        self._ck(self.L.smi_dev_fri_fold(self.h, vp(d_in), length, vp(d_alpha), offset, omega, vp(d_out)))

    def dev_fri_fold_shard(self, d_lo, d_hi, count, index0, full_len, d_alpha, offset, omega, d_out):
        self._ck(self.L.smi_dev_fri_fold_shard(self.h, vp(d_lo), vp(d_hi), count, index0, full_len, vp(d_alpha), offset, omega,
                                               vp(d_out)))

    def dev_fri_prove(self, cfg, d_codeword, length):
        proof, plen = vp(), C.c_size_t()
        top = np.zeros(max(cfg.num_colinearity_tests, 1), dtype=np.uint64)
        self._ck(self.L.smi_dev_fri_prove(self.h, C.byref(cfg), vp(d_codeword), length, C.byref(proof), C.byref(plen),
                                          top.ctypes.data, None))
        b = C.string_at(proof, plen.value)
        self.L.smi_free(proof)
        return b, [int(v) for v in top[:cfg.num_colinearity_tests]]

    def dev_combine_columns(self, d_cols, n_cols, length, stride, d_weights, d_out):
        self._ck(self.L.smi_dev_combine_columns(self.h, vp(d_cols), n_cols, length, stride, vp(d_weights), vp(d_out)))

    def dev_stark_prove(self, d_trace_cols, n_cols, log_n, log_blowup, num_colinearity_tests, trace_offset=1,
                        lde_offset=None, timed=False, row_leaves=False, open_columns=False):
        """Build-defined prove.  -> dict(column_roots, proof, top_indices[, stage_ms]).
        row_leaves: commit to the extended trace with one tree over its rows instead of one per column."""
        cfg = _lib.StarkCfg(log_n, log_blowup, n_cols, 1 if row_leaves else 0, trace_offset,
                            self.g if lde_offset is None else lde_offset, num_colinearity_tests, 1 if open_columns else 0)
        roots = np.zeros((1 if row_leaves else n_cols, 32), dtype=np.uint8)
        proof, plen = vp(), C.c_size_t()
        top = np.zeros(max(num_colinearity_tests, 1), dtype=np.uint64)
        stage = (C.c_double * 4)()
        self._ck(self.L.smi_dev_stark_prove(self.h, C.byref(cfg), vp(d_trace_cols), roots.ctypes.data, C.byref(proof),
                                            C.byref(plen), top.ctypes.data, stage if timed else None))
        b = C.string_at(proof, plen.value)
        self.L.smi_free(proof)
        out = {"column_roots": roots, "proof": b, "top_indices": [int(v) for v in top[:num_colinearity_tests]]}
        if timed:
            out["stage_ms"] = dict(zip(("lde", "commit", "combine", "fri"), [float(x) for x in stage]))
        return out


class DeviceTree:
    """MerkleTree kept on the device (all levels, src/merkle.rs:4-8)."""

    def __init__(self, eng, handle):
        self.eng, self.h = eng, handle
        self.n = int(eng.L.smi_merkle_num_leaves(handle))

    def root(self) -> bytes:
        out = (C.c_uint8 * 32)()
        self.eng._ck(self.eng.L.smi_merkle_root(self.eng.h, self.h, out))
        return bytes(out)

    def open(self, index):
        path = np.zeros((64, 32), dtype=np.uint8)
        depth = C.c_size_t()
        self.eng._ck(self.eng.L.smi_merkle_open(self.eng.h, self.h, index, path.ctypes.data, C.byref(depth)))
        return [bytes(path[i]) for i in range(depth.value)]

    def level(self, lvl):
        out = np.zeros((max(self.n >> lvl, 1), 32), dtype=np.uint8)
        cnt = C.c_size_t()
        self.eng._ck(self.eng.L.smi_merkle_level(self.eng.h, self.h, lvl, out.ctypes.data, C.byref(cnt)))
        return out[:cnt.value]

    def free(self):
        if self.h:
            self.eng.L.smi_merkle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FriRun:
    """Device-resident result of Fri::commit: the codewords it returns and the per-round trees."""

    def __init__(self, eng, handle):
        self.eng, self.h = eng, handle

    def __len__(self):
        n = C.c_size_t()
        self.eng._ck(self.eng.L.smi_fri_run_num_codewords(self.h, C.byref(n)))
        return n.value

    def codeword(self, rnd):
        ln = C.c_size_t()
        self.eng._ck(self.eng.L.smi_fri_run_codeword(self.h, rnd, None, C.byref(ln)))
        out = np.empty(ln.value, dtype=np.uint64)
        self.eng._ck(self.eng.L.smi_fri_run_codeword(self.h, rnd, out.ctypes.data, C.byref(ln)))
        return out

    def open(self, rnd, index):
        path = np.zeros((64, 32), dtype=np.uint8)
        depth = C.c_size_t()
        self.eng._ck(self.eng.L.smi_fri_run_open(self.h, rnd, index, path.ctypes.data, C.byref(depth)))
        return [bytes(path[i]) for i in range(depth.value)]

    def free(self):
        if self.h:
            self.eng.L.smi_fri_run_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def default_engine(p=P_REF, g=G_REF, device=0):
    """Process-wide engine per (p, device); created on first use."""
    key = (p, device)
    if key not in _default:
        _default[key] = Engine(p, g, device)
    return _default[key]
