"""Multi-GPU prover: thin ctypes caller of the smi_mgpu_* entry points (include/stark_mi.h,
csrc/mgpu.hip).  One process per GPU; the round loop, the kernels and the RCCL collectives all
run inside libstarkmi.so on the engine's stream -- this module only creates the communicator
and passes pointers.

MultiGpu(engine, rank, world)              RCCL over xGMI: rank 0 draws the unique id, `carry`
                                           (default: torch.distributed broadcast_object_list)
                                           takes it to the other ranks.
MultiGpu(engine, rank, world, host=...)    the same prover over a caller-supplied collective shim
                                           (HostCollectives over gloo): how the multi-rank logic is
                                           exercised on a one-GPU box.

HostCollectives is also what tests/test_mgpu_gloo.py hands to the CPU instantiation of the same
loop (libstarkmi_emu.so, test infrastructure)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FriCfg, StarkCfg, check

vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int

ALL_GATHER = C.CFUNCTYPE(i32, vp, vp, vp, sz)
EXCHANGE = C.CFUNCTYPE(i32, vp, i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(sz), i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(sz))
ALL_REDUCE = C.CFUNCTYPE(i32, vp, vp, sz)


class CollOps(C.Structure):
    """smi_mgpu_coll"""
    _fields_ = [("user", vp), ("all_gather", ALL_GATHER), ("exchange", EXCHANGE), ("all_reduce_sum_u8", ALL_REDUCE)]


class HostMem:
    """pointers are host memory (the CPU instantiation of the loop)"""

    def read(self, ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, np.uint8)

    def write(self, ptr, arr):
        if arr.size:
            C.memmove(ptr, arr.ctypes.data, arr.size)


class HipMem:
    """pointers are device memory: staged through the host with hipMemcpy (rehearsals only -- the
    product path is RCCL, device to device)"""

    def __init__(self):
        import torch  # noqa: F401  (one HIP runtime per process: the copy torch loaded)
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [vp, vp, sz, i32]

    def read(self, ptr, n):
        out = np.empty(n, dtype=np.uint8)
        if n and self.hip.hipMemcpy(out.ctypes.data, ptr, n, 2):
            raise RuntimeError("hipMemcpy D2H failed")
        return out

    def write(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        if arr.size and self.hip.hipMemcpy(ptr, arr.ctypes.data, arr.size, 1):
            raise RuntimeError("hipMemcpy H2D failed")


class HostCollectives:
    """The three collectives of smi_mgpu_coll over torch.distributed on CPU tensors (gloo)."""

    def __init__(self, rank, world, mem, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.mem, self.group = torch, dist, rank, world, mem, group
        self.errors = []
        self._keep = (ALL_GATHER(self._all_gather), EXCHANGE(self._exchange), ALL_REDUCE(self._all_reduce))
        self.ops = CollOps(None, *self._keep)

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:      # never let an exception cross the C frame
            self.errors.append(repr(e))
            return 1

    def _all_gather(self, _user, send, recv, n):
        def run():
            t = self.torch.from_numpy(self.mem.read(send, n))
            outs = [self.torch.empty(n, dtype=self.torch.uint8) for _ in range(self.world)]
            self.dist.all_gather(outs, t, group=self.group)
            self.mem.write(recv, self.torch.cat(outs).numpy())
        return self._guard(run)

    def _exchange(self, _user, n_send, send_peer, send_ptr, send_bytes, n_recv, recv_peer, recv_ptr, recv_bytes):
        def run():
            ops, bufs = [], []
            for k in range(n_send):
                t = self.torch.from_numpy(self.mem.read(send_ptr[k], send_bytes[k]))
                ops.append(self.dist.P2POp(self.dist.isend, t, send_peer[k], group=self.group))
            for k in range(n_recv):
                b = self.torch.empty(recv_bytes[k], dtype=self.torch.uint8)
                bufs.append(b)
                ops.append(self.dist.P2POp(self.dist.irecv, b, recv_peer[k], group=self.group))
            if ops:
                for r in self.dist.batch_isend_irecv(ops):
                    r.wait()
            for k in range(n_recv):
                self.mem.write(recv_ptr[k], bufs[k].numpy())
        return self._guard(run)

    def _all_reduce(self, _user, buf, n):
        def run():
            t = self.torch.from_numpy(self.mem.read(buf, n))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.mem.write(buf, t.numpy())
        return self._guard(run)


def _sig(L):
    if getattr(L, "_mgpu_sig", False):
        return
    L.smi_mgpu_unique_id.argtypes = [vp]
    L.smi_mgpu_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.smi_mgpu_create_with.argtypes = [vp, C.POINTER(CollOps), i32, i32, C.POINTER(vp)]
    L.smi_mgpu_destroy.argtypes = [vp]
    L.smi_mgpu_destroy.restype = None
    L.smi_mgpu_set_min_block.argtypes = [vp, sz]
    L.smi_mgpu_fri_commit.argtypes = [vp, C.POINTER(FriCfg), vp, sz, vp, vp, vp, C.POINTER(sz)]
    L.smi_mgpu_fri_prove.argtypes = [vp, C.POINTER(FriCfg), vp, sz, C.POINTER(vp), C.POINTER(sz), vp]
    L.smi_mgpu_lde.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    L.smi_mgpu_stark_prove.argtypes = [vp, C.POINTER(StarkCfg), vp, vp, C.POINTER(vp), C.POINTER(sz), vp]
    L.smi_mgpu_ntt.argtypes = [vp, vp, vp, C.c_uint32, i32, C.c_uint64]
    L.smi_mgpu_ntt_natural.argtypes = [vp, vp, vp, C.c_uint32, i32, C.c_uint64]
    L.smi_mgpu_ntt_first_digit.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
    L._mgpu_sig = True


def torch_carry(id_bytes, rank, group=None):
    """rank 0's unique id to every rank over the default torch.distributed group (any backend)."""
    import torch.distributed as dist
    box = [id_bytes if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


class MultiGpu:
    def __init__(self, engine, rank, world, host=None, carry=torch_carry, min_block=None):
        self.eng, self.rank, self.world, self.host = engine, rank, world, host
        self.L = _lib.lib()
        _sig(self.L)
        h = vp()
        if host is not None:
            check(self.L.smi_mgpu_create_with(engine.h, C.byref(host.ops), rank, world, C.byref(h)), engine.h)
        else:
            buf = (C.c_uint8 * 128)()
            st = self.L.smi_mgpu_unique_id(buf) if rank == 0 else 0
            # rank 0 always takes part in the carry, so a failure there cannot leave the others waiting
            ident = bytes(buf) if st == 0 else b""
            if world > 1:
                ident = carry(ident, rank)
            check(st, engine.h)
            if len(ident) != 128:
                raise RuntimeError("rank 0 could not draw an RCCL unique id")
            buf = (C.c_uint8 * 128).from_buffer_copy(ident)
            check(self.L.smi_mgpu_create(engine.h, buf, rank, world, C.byref(h)), engine.h)
        self.h = h
        if min_block is not None:
            check(self.L.smi_mgpu_set_min_block(self.h, min_block), engine.h)

    def close(self):
        if self.h:
            self.L.smi_mgpu_destroy(self.h)
            self.h = None

    def _ck(self, st):
        if st and self.host is not None and self.host.errors:
            raise RuntimeError("collective shim: " + "; ".join(self.host.errors))
        check(st, self.eng.h)

    def _take(self, proof, plen):
        out = C.string_at(proof, plen.value)
        self.L.smi_free(proof)
        return out

    def fri_commit(self, cfg, d_block, block_len):
        """-> (roots [R x bytes], alphas [R-1 ints], last codeword np.uint64) on every rank"""
        rounds = C.c_uint64()
        check(self.L.smi_fri_num_rounds(C.byref(cfg), C.byref(rounds)))
        R = max(rounds.value, 1)
        roots = (C.c_uint8 * (32 * R))()
        alphas = (C.c_uint64 * R)()
        last = np.zeros(cfg.domain_length, dtype=np.uint64)
        n = sz()
        self._ck(self.L.smi_mgpu_fri_commit(self.h, C.byref(cfg), d_block, block_len, roots, alphas, last.ctypes.data, C.byref(n)))
        raw = bytes(roots)
        return [raw[32 * i:32 * i + 32] for i in range(rounds.value)], list(alphas)[:rounds.value - 1], last[:n.value]

    def fri_prove(self, cfg, d_block, block_len):
        """-> (serialized ProofStream, top-level indices) on every rank"""
        proof, plen = vp(), sz()
        top = (C.c_uint64 * max(int(cfg.num_colinearity_tests), 1))()
        self._ck(self.L.smi_mgpu_fri_prove(self.h, C.byref(cfg), d_block, block_len, C.byref(proof), C.byref(plen), top))
        return self._take(proof, plen), list(top)[:int(cfg.num_colinearity_tests)]

    def lde(self, d_trace_cols, n_cols, log_n, log_blowup, d_out_blocks, trace_offset=1, lde_offset=None):
        lde_offset = self.eng.g if lde_offset is None else lde_offset
        self._ck(self.L.smi_mgpu_lde(self.h, d_trace_cols, n_cols, log_n, log_blowup, trace_offset, lde_offset, d_out_blocks))

    def ntt_first_digit(self, log_n):
        r0 = C.c_uint32()
        check(self.L.smi_mgpu_ntt_first_digit(log_n, C.byref(r0)))
        return r0.value

    def ntt(self, d_strip, d_out, log_n, inverse=False, offset=1, natural=False):
        """one 2^log_n-point transform over the ranks (layouts: include/stark_mi.h, smi_mgpu_ntt); natural: the result
        as this rank's contiguous natural-order block (smi_mgpu_ntt_natural, one more all-to-all)"""
        fn = self.L.smi_mgpu_ntt_natural if natural else self.L.smi_mgpu_ntt
        self._ck(fn(self.h, d_strip, d_out, log_n, 1 if inverse else 0, offset))

    def stark_prove(self, d_trace_cols, n_cols, log_n, log_blowup, num_colinearity_tests, trace_offset=1, lde_offset=None,
                    open_columns=False):
        """-> (column roots [W x bytes], proof bytes, top-level indices) on every rank"""
        lde_offset = self.eng.g if lde_offset is None else lde_offset
        cfg = StarkCfg(log_n, log_blowup, n_cols, 0, trace_offset, lde_offset, num_colinearity_tests, 1 if open_columns else 0)
        roots = (C.c_uint8 * (32 * n_cols))()
        proof, plen = vp(), sz()
        top = (C.c_uint64 * max(num_colinearity_tests, 1))()
        self._ck(self.L.smi_mgpu_stark_prove(self.h, C.byref(cfg), d_trace_cols, roots, C.byref(proof), C.byref(plen), top))
        raw = bytes(roots)
        return [raw[32 * i:32 * i + 32] for i in range(n_cols)], self._take(proof, plen), list(top)[:num_colinearity_tests]
