"""GPU: the multi-GPU entry points of the C ABI (smi_mgpu_*, csrc/mgpu.hip).

  * world size 1 over a real RCCL communicator: the code path the 8-GPU runs take (ncclCommInitRank,
    collectives on the engine's stream) must give the single-GPU entry points' bytes;
  * world sizes 2 and 4 with every rank on THIS box's one GPU: the same HIP kernels and round loop,
    the collectives through the smi_mgpu_coll shim (gloo, staged through the host).  RCCL refuses
    several ranks on one device, so this is how owner-writes / all-reduce assembly / fold exchange /
    coset all-to-all are exercised with real kernels before the driver's 8-GPU run.
`pytest -m gpu`."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P, G = 998244353, 3
P2, G2 = 469762049, 3


def _vals(o, seed, n, p):
    return o.splitmix64(seed, n) % np.uint64(p)


def _upload(eng, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    d = eng.dev_alloc(arr.size * 4)
    eng.dev_upload(arr, d)
    return d


@pytest.mark.parametrize("p,g", [(P, G), (P2, G2)])
def test_rccl_world1_equals_single_gpu_entry_points(oracle, p, g):
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import MultiGpu
    o = oracle
    eng = s.Engine(p, g, 0)
    mg = MultiGpu(eng, 0, 1)           # ncclCommInitRank with one rank
    logn, lb, W, t = 12, 3, 4, 8
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, p) for c in range(W)])
    d = _upload(eng, cols)
    want = eng.dev_stark_prove(d, W, logn, lb, t)
    roots, proof, top = mg.stark_prove(d, W, logn, lb, t)
    assert roots == [bytes(r) for r in want["column_roots"]]
    assert proof == want["proof"] and top == want["top_indices"]
    # Fri::prove / Fri::commit on their own, against the oracle
    omega, offset = eng.prim_nth_root(N), 3
    cw = o.fast_coset_ntt(_vals(o, 6, N // 8, p), N, omega, offset, p)
    d_cw = _upload(eng, cw)
    cfg = eng.fri_cfg(omega, offset, N, 8, t)
    ocfg = o.fri_cfg(omega, offset, N, 8, t, p)
    got, gtop = mg.fri_prove(cfg, d_cw, N)
    wantp, wtop = o.fri_prove(ocfg, cw)
    assert got == wantp and gtop == wtop
    r2, a2, last = mg.fri_commit(cfg, d_cw, N)
    wr, wa, wl = o.fri_commit_trace(ocfg, cw)
    assert r2 == [bytes(r) for r in wr] and a2 == wa and np.array_equal(last, wl)
    eng.dev_free(d)
    eng.dev_free(d_cw)
    mg.close()
    eng.close()


def _rank_main(rank, world, port, p, g, logn, lb, W, t, min_block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import HipMem, HostCollectives, MultiGpu
    from oracle import oracle as o
    o.build()
    eng = s.Engine(p, g, 0)                                   # every rank on the box's one GPU
    coll = HostCollectives(rank, world, HipMem())
    mg = MultiGpu(eng, rank, world, host=coll, min_block=min_block)
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, p) for c in range(W)])
    d = _upload(eng, cols)
    ok = True
    # the sharded extension: this rank's block of every column == the same block of smi_dev_lde
    blk = N // world
    d_full = eng.dev_alloc(W * N * 4)
    eng.dev_lde(d, W, logn, lb, d_full, 1, g)
    full = eng.dev_download(d_full, W * N).reshape(W, N)
    d_blk = eng.dev_alloc(W * blk * 4)
    mg.lde(d, W, logn, lb, d_blk, 1, g)
    got = eng.dev_download(d_blk, W * blk).reshape(W, blk)
    ok = ok and np.array_equal(got, full[:, rank * blk:(rank + 1) * blk])
    # the whole prove: every rank must hold the single-GPU bytes
    want = eng.dev_stark_prove(d, W, logn, lb, t)
    roots, proof, top = mg.stark_prove(d, W, logn, lb, t)
    ok = ok and roots == [bytes(r) for r in want["column_roots"]] and proof == want["proof"] and top == want["top_indices"]
    # Fri::prove of a sharded codeword against the oracle
    omega, offset = eng.prim_nth_root(N), 3
    cw = o.fast_coset_ntt(_vals(o, 6, N // 8, p), N, omega, offset, p)
    d_cw = _upload(eng, cw[rank * blk:(rank + 1) * blk])
    got, gtop = mg.fri_prove(eng.fri_cfg(omega, offset, N, 8, t), d_cw, blk)
    wantp, wtop = o.fri_prove(o.fri_cfg(omega, offset, N, 8, t, p), cw)
    ok = ok and got == wantp and gtop == wtop
    q.put((rank, bool(ok), coll.errors))
    mg.close()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,p,g,logn,lb,W,t,min_block", [
    (2, P, G, 13, 3, 4, 8, 1 << 10),     # two whole columns per rank, FRI sharded down to 2^10 blocks
    (4, P2, G2, 14, 3, 4, 8, 1 << 9),    # one column per rank (8 cosets each), second prime
])
def test_two_and_four_ranks_on_one_gpu_through_the_collective_shim(oracle, world, p, g, logn, lb, W, t, min_block):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, p, g, logn, lb, W, t, min_block, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(420)
        assert pr.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(g_[1] for g_ in got), got


def test_sharded_ntt_world1_at_2p26_is_the_direct_transform(oracle):
    """BASELINE configs[3] at full size on the one GPU of the box: smi_mgpu_ntt (the pass pipeline with
    the exchange as the identity) over a real RCCL communicator.  2^26 points on the second prime:
    equal to smi_dev_ntt's output (same passes), the op-for-op oracle's Polynomial::eval at sampled
    points (src/univariate/eval.rs:6-14) and a forward/inverse round trip."""
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import MultiGpu
    o = oracle
    eng = s.Engine(P2, G2, 0)
    mg = MultiGpu(eng, 0, 1)
    L = 26
    n = 1 << L
    coef_n = 1 << 10                                          # a sparse input keeps the oracle's power sums cheap
    x = np.zeros(n, dtype=np.uint64)
    x[:coef_n] = _vals(o, 4, coef_n, P2)
    d_in, d_out, d_ref = _upload(eng, x), eng.dev_alloc(n * 4), eng.dev_alloc(n * 4)
    eng.dev_ntt(d_in, d_ref, L, offset=3)
    mg.ntt(d_in, d_out, L, offset=3)                          # d_in is clobbered
    got = eng.dev_download(d_out, n)
    assert np.array_equal(got, eng.dev_download(d_ref, n))
    w = eng.prim_nth_root(n)
    for k in (0, 1, 12345, n // 2 + 7, n - 1):
        assert int(got[k]) == o.poly_eval(x[:coef_n], o.ff_mul(3, o.ff_exp(w, k, P2), P2), P2)
    # round trip: values on the subgroup -> coefficients -> values
    vals = _vals(o, 9, n, P2)
    eng.dev_upload(vals, d_in)
    mg.ntt(d_in, d_out, L, inverse=True)
    mg.ntt(d_out, d_in, L)
    assert np.array_equal(eng.dev_download(d_in, n), vals)
    for d in (d_in, d_out, d_ref):
        eng.dev_free(d)
    mg.close()
    eng.close()


def _ntt_rank_main(rank, world, port, logn, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import HipMem, HostCollectives, MultiGpu
    from oracle import oracle as o
    from test_mgpu_gloo import shard_strip, unshard_output
    o.build()
    eng = s.Engine(P2, G2, 0)
    coll = HostCollectives(rank, world, HipMem())
    mg = MultiGpu(eng, rank, world, host=coll)
    n = 1 << logn
    x = _vals(o, 5 + logn, n, P2)
    r0 = mg.ntt_first_digit(logn)
    d_strip = _upload(eng, shard_strip(x, r0, rank, world))
    d_out = eng.dev_alloc((n // world) * 4)
    mg.ntt(d_strip, d_out, logn, offset=7)
    mine = eng.dev_download(d_out, n // world).astype(np.uint32)
    parts = [torch.zeros(n // world, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(mine.view(np.int32)))
    got = unshard_output([t.numpy().view(np.uint32) for t in parts], r0, world).astype(np.uint64)
    want = o.fast_coset_ntt(x, n, o.ff_prim_nth_root_g(n, P2, G2), 7, P2)
    q.put((rank, bool(np.array_equal(got, want)), coll.errors))
    mg.close()
    eng.close()
    dist.destroy_process_group()


def _ntt_commit_rank_main(rank, world, port, logn, q):
    """smi_mgpu_ntt_natural -> smi_mgpu_fri_prove on the device: a low-degree polynomial evaluated by ONE transform over
    the ranks, each rank's contiguous natural-order block handed to the sharded Fri::prove without a host copy."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import HipMem, HostCollectives, MultiGpu
    from oracle import oracle as o
    from test_mgpu_gloo import shard_strip
    o.build()
    eng = s.Engine(P2, G2, 0)
    coll = HostCollectives(rank, world, HipMem())
    mg = MultiGpu(eng, rank, world, host=coll, min_block=1 << 12)
    n, exp, t, offset = 1 << logn, 8, 8, 5
    coeffs = np.zeros(n, dtype=np.uint64)
    coeffs[:n // exp] = _vals(o, 17 + logn, n // exp, P2)
    r0 = mg.ntt_first_digit(logn)
    d_strip = _upload(eng, shard_strip(coeffs, r0, rank, world))
    blk = n // world
    d_block = eng.dev_alloc(blk * 4)
    mg.ntt(d_strip, d_block, logn, offset=offset, natural=True)
    omega = eng.prim_nth_root(n)
    codeword = o.fast_coset_ntt(coeffs[:n // exp], n, omega, offset, P2)
    ok = np.array_equal(eng.dev_download(d_block, blk), codeword[rank * blk:(rank + 1) * blk])
    proof, top = mg.fri_prove(eng.fri_cfg(omega, offset, n, exp, t), d_block, blk)
    d_all = _upload(eng, codeword)
    want, want_top = eng.dev_fri_prove(eng.fri_cfg(omega, offset, n, exp, t), d_all, n)      # the single-GPU proof (itself == oracle elsewhere)
    ok = ok and proof == bytes(want) and top == list(want_top)
    ok = ok and (rank != 0 or o.fri_verify(o.fri_cfg(omega, offset, n, exp, t, P2), proof))
    q.put((rank, bool(ok), coll.errors))
    mg.close()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,logn", [(2, 20), (4, 22)])
def test_natural_order_transform_feeds_the_sharded_prove_on_one_gpu(oracle, world, logn):
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_ntt_commit_rank_main, args=(r, world, port, logn, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(420)
        assert pr.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(g_[1] for g_ in got), got


@pytest.mark.parametrize("world,logn", [(2, 22), (4, 24)])
def test_sharded_ntt_ranks_on_one_gpu(oracle, world, logn):
    """smi_mgpu_ntt with 2 and 4 ranks on the box's one GPU (collectives through the gloo shim): strips in,
    ONE exchange, natural-order runs out -- the whole coset transform against the oracle."""
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_ntt_rank_main, args=(r, world, port, logn, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(420)
        assert pr.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(g_[1] for g_ in got), got


def _rccl_rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)      # only carries the 128-byte communicator id
    import stark_rs_amd as s
    from stark_rs_amd.mgpu import MultiGpu
    from oracle import oracle as o
    o.build()
    eng = s.Engine(P2, G2, rank)                                      # one GPU per rank
    mg = MultiGpu(eng, rank, world, min_block=1 << 10)                # RCCL communicator over the ranks' GPUs
    logn, lb, W, t = 14, 3, 4, 8
    n, N = 1 << logn, 1 << (logn + lb)
    cols = np.stack([_vals(o, 0x5354524B00 + c, n, P2) for c in range(W)])
    d = _upload(eng, cols)
    want = eng.dev_stark_prove(d, W, logn, lb, t, open_columns=True)
    roots, proof, top = mg.stark_prove(d, W, logn, lb, t, open_columns=True)
    ok = roots == [bytes(r) for r in want["column_roots"]] and proof == want["proof"] and top == want["top_indices"]
    # the sharded extension's block and the natural-order transform's block against the single-GPU results
    d_ref, d_blk = eng.dev_alloc(W * N * 4), eng.dev_alloc(W * (N // world) * 4)
    eng.dev_lde(d, W, logn, lb, d_ref)
    mg.lde(d, W, logn, lb, d_blk)
    ref = eng.dev_download(d_ref, W * N).reshape(W, N)
    ok = ok and np.array_equal(eng.dev_download(d_blk, W * (N // world)).reshape(W, N // world), ref[:, rank * (N // world):(rank + 1) * (N // world)])
    from test_mgpu_gloo import shard_strip
    L = 20
    x = _vals(o, 3, 1 << L, P2)
    d_x, d_y = _upload(eng, x), eng.dev_alloc(4 << L)
    eng.dev_ntt(d_x, d_y, L, offset=7)
    d_strip, d_nat = _upload(eng, shard_strip(x, mg.ntt_first_digit(L), rank, world)), eng.dev_alloc((4 << L) // world)
    mg.ntt(d_strip, d_nat, L, offset=7, natural=True)
    per = (1 << L) // world
    ok = ok and np.array_equal(eng.dev_download(d_nat, per), eng.dev_download(d_y, 1 << L)[rank * per:(rank + 1) * per])
    q.put((rank, bool(ok)))
    mg.close()
    eng.close()
    dist.destroy_process_group()


def test_rccl_over_several_gpus_when_the_box_has_them(oracle):
    """Runs only on a multi-GPU box (the development boxes have one): 2 or 4 ranks, one GPU each, the real
    RCCL collectives over xGMI -- every rank must hold the single-GPU proof bytes."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("one GPU on this box: the multi-rank logic runs through the collective shim instead")
    world = 4 if ndev >= 4 else 2
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_rccl_rank_main, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(420)
        assert pr.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(g_[1] for g_ in got), got
