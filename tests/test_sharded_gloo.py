"""CPU, world_size 2 and 4 over gloo: Fri::commit and Fri::prove of one codeword sharded over ranks
(stark_rs_amd/sharded.py) -- per-rank Merkle subtrees + all-gather of sub-roots, replicated
Fiat-Shamir, perfect-shuffle fold exchange, final gather -- against the oracle's Fri::commit.
Local steps run the kernels' own arithmetic through the CPU emulator (hash_core.h, fri_core.h)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

P, G = 998244353, 3
u32p = C.POINTER(C.c_uint32)


class EmuShardBackend:
    def __init__(self, p, g):
        import stark_rs_amd as s
        s.build()
        self.L = C.CDLL(os.path.join(os.path.dirname(s.__file__), "build", "libstarkmi_emu.so"))
        self.L.emu_fold_shard.argtypes = [C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                          C.c_uint64, C.c_uint64, u32p]
        self.p, self.g = p, g

    def tensor(self, values):
        return torch.from_numpy(np.ascontiguousarray(values, dtype=np.uint32).view(np.int32).copy())

    class Tree:
        def __init__(self, be, cw):
            v = np.ascontiguousarray(cw.numpy().view(np.uint32))
            d = np.zeros((len(v), 32), dtype=np.uint8)
            be.L.emu_leaf_hash(v.ctypes.data_as(C.c_void_p), C.c_size_t(len(v)), d.ctypes.data_as(C.c_void_p))
            self.n, self.levels = len(v), [d]
            while len(self.levels[-1]) > 1:
                self.levels.append(be.hash_pairs(self.levels[-1]))
            self.root = bytes(self.levels[-1][0])

        def open_many(self, indices):
            return [[bytes(lv[(i >> l) ^ 1]) for l, lv in enumerate(self.levels[:-1])] for i in indices]

    def subtree(self, cw):
        return EmuShardBackend.Tree(self, cw)

    def values(self, cw, indices):
        v = cw.numpy().view(np.uint32)
        return [int(v[i]) for i in indices]

    def hash_pairs(self, digests):
        digests = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros((len(digests) // 2, 32), dtype=np.uint8)
        pairs = np.ascontiguousarray(digests.reshape(-1, 64))
        self.L.emu_node_hash(pairs.ctypes.data_as(C.c_void_p), C.c_size_t(len(out)), out.ctypes.data_as(C.c_void_p))
        return out

    def hash_bytes(self, data):
        out = (C.c_uint8 * 32)()
        self.L.emu_hash_bytes(data, C.c_size_t(len(data)), out)
        return bytes(out)

    def fold(self, lo, hi, index0, full_len, alpha, offset, omega):
        a = np.ascontiguousarray(lo.numpy().view(np.uint32)); b = np.ascontiguousarray(hi.numpy().view(np.uint32))
        out = np.zeros(len(a), dtype=np.uint32)
        rc = self.L.emu_fold_shard(self.p, self.g, a.ctypes.data_as(u32p), b.ctypes.data_as(u32p), len(a), index0, full_len,
                                   alpha, offset, omega, out.ctypes.data_as(u32p))
        assert rc == 0
        return torch.from_numpy(out.view(np.int32))

    def fence(self):
        pass

    def lde(self, trace, n_cols, log_n, log_blowup, trace_offset, lde_offset):
        from oracle import oracle as o
        n, N = 1 << log_n, 1 << (log_n + log_blowup)
        w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
        cols = trace.numpy().view(np.uint32).astype(np.uint64).reshape(n_cols, n)
        out = np.concatenate([o.fast_coset_ntt(o.fast_intt(cols[c], w, trace_offset), N, Wn, lde_offset) for c in range(n_cols)])
        return self.tensor(out)

    def combine(self, cols, n_cols, stride, start, length, weights):
        v = cols.numpy().view(np.uint32).astype(object)
        acc = np.zeros(length, dtype=object)
        for c in range(n_cols):
            acc = (acc + (weights[c] % self.p) * v[c * stride + start:c * stride + start + length]) % self.p
        return self.tensor(acc.astype(np.uint64))


def _worker(rank, world, port, logn, expansion, t, offset, min_block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stark_rs_amd.sharded import ShardedFriCommit
    from oracle import oracle as o
    n = 1 << logn
    omega = o.ff_prim_nth_root(n)
    coeffs = o.splitmix64(77, n // expansion) % np.uint64(P)
    codeword = o.fast_coset_ntt(coeffs, n, omega, offset)
    be = EmuShardBackend(P, G)
    blk = n // world
    fc = ShardedFriCommit(be, P, omega, offset, n, expansion, t, rank, world, min_block=min_block)
    roots, alphas, last = fc.commit(be.tensor(codeword[rank * blk:(rank + 1) * blk]))
    if rank == 0:
        cfg = o.fri_cfg(omega, offset, n, expansion, t)
        wroots, walphas, wlast = o.fri_commit_trace(cfg, codeword)
        ok = [bytes(r) for r in wroots] == roots and walphas == alphas and \
            np.array_equal(last.numpy().view(np.uint32).astype(np.uint64), wlast)
        q.put(bool(ok))
    dist.destroy_process_group()


def _prove_worker(rank, world, port, logn, expansion, t, offset, min_block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stark_rs_amd.sharded import ShardedFriProve
    from oracle import oracle as o
    n = 1 << logn
    omega = o.ff_prim_nth_root(n)
    coeffs = o.splitmix64(78, n // expansion) % np.uint64(P)
    codeword = o.fast_coset_ntt(coeffs, n, omega, offset)
    be = EmuShardBackend(P, G)
    blk = n // world
    fp = ShardedFriProve(be, P, omega, offset, n, expansion, t, rank, world, min_block=min_block)
    proof, top = fp.prove(be.tensor(codeword[rank * blk:(rank + 1) * blk]))
    if rank == 0:
        cfg = o.fri_cfg(omega, offset, n, expansion, t)
        want, want_top = o.fri_prove(cfg, codeword)
        q.put(bool(proof == want and top == want_top and o.fri_verify(cfg, proof)))
    dist.destroy_process_group()


def _stark_worker(rank, world, port, logn, lb, W, t, min_block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stark_rs_amd.sharded import ShardedStarkProve
    from oracle import oracle as o
    n, N = 1 << logn, 1 << (logn + lb)
    w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    cols = np.stack([o.splitmix64(0x5354524B00 + c, n) % np.uint64(P) for c in range(W)])
    be = EmuShardBackend(P, G)
    sp = ShardedStarkProve(be, P, G, logn, lb, W, t, Wn, rank, world, min_block=min_block)
    roots, proof, top = sp.prove(be.tensor(cols.reshape(-1)))
    if rank == 0:
        # the single-process composition, stage by stage (what tests/test_gpu_pipeline.py checks the
        # one-GPU smi_dev_stark_prove against)
        lde = [o.fast_coset_ntt(o.fast_intt(cols[c], w, 1), N, Wn, G) for c in range(W)]
        fs, weights, want_roots = o.FiatShamir(), [], []
        for c in range(W):
            want_roots.append(o.merkle_commit(o.leaf_hashes(lde[c])))
            fs.absorb(want_roots[-1])
            weights.append(fs.challenge() % P)
        cw = np.zeros(N, dtype=object)
        for c in range(W):
            cw = (cw + lde[c].astype(object) * weights[c]) % P
        cfg = o.fri_cfg(Wn, G, N, 1 << lb, t)
        want, want_top = o.fri_prove(cfg, cw.astype(np.uint64))
        q.put(bool(roots == [bytes(r) for r in want_roots] and proof == want and top == want_top and o.fri_verify(cfg, proof)))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,logn,expansion,t,offset,min_block", [
    (2, 10, 4, 4, 3, 64),       # sharded for 3 rounds, then gathered
    (4, 11, 8, 8, 7, 32),       # four ranks: lo/hi partners differ
    (2, 8, 4, 2, 3, 1 << 12),   # block below min_block from the start: gathered immediately
])
def test_sharded_fri_commit_gloo(oracle, world, logn, expansion, t, offset, min_block):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, logn, expansion, t, offset, min_block, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world,logn,expansion,t,offset,min_block", [
    (2, 10, 4, 4, 3, 64),       # openings come from both ranks' subtrees, later rounds replicated
    (4, 11, 8, 8, 7, 32),       # two levels above the sub-roots
    (2, 8, 4, 2, 3, 1 << 12),   # gathered from the start: rank 0 answers every query
])
def test_sharded_fri_prove_is_byte_identical_gloo(oracle, world, logn, expansion, t, offset, min_block):
    """Fri::prove over a sharded codeword: the serialized proof must be byte for byte the oracle's
    single-process proof, and the oracle's Fri::verify accepts it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prove_worker, args=(r, world, port, logn, expansion, t, offset, min_block, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world,logn,lb,W,t,min_block", [
    (2, 8, 3, 4, 4, 64),        # column subtrees on two ranks, FRI sharded for a few rounds
    (4, 7, 2, 3, 2, 16),        # three columns on four ranks: two levels above the sub-roots
])
def test_sharded_stark_prove_equals_single_process_composition_gloo(oracle, world, logn, lb, W, t, min_block):
    """ShardedStarkProve (replicated LDE, sharded column trees and FRI): column roots, proof bytes and
    top-level indices equal the single-process composition of csrc/stark.hip restated with the oracle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stark_worker, args=(r, world, port, logn, lb, W, t, min_block, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True
