#!/usr/bin/env python3
"""Development tool (one-GPU box): ShardedFriProve with the HIP backend at world size > 1, all ranks on
cuda:0 over gloo, checked byte for byte against the single-GPU smi_fri_prove.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29512 tools/rehearse_sharded.py 25
(bench.py has the matching knobs SMI_BENCH_BACKEND=gloo SMI_BENCH_DEVICE=0 for its N > 1 legs.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import stark_rs_amd as s  # noqa: E402
from bench import splitmix64  # noqa: E402
from stark_rs_amd.sharded import HipShardBackend, ShardedFriProve  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    eng = s.Engine(s.P2, s.G2, 0)
    logN = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    N, p = 1 << logN, s.P2
    full = np.concatenate([(splitmix64(9 + r, N // world) % np.uint64(p)).astype(np.uint32) for r in range(world)])
    blk = N // world
    be = HipShardBackend(eng)
    fp = ShardedFriProve(be, p, eng.prim_nth_root(N), s.G2, N, 8, 32, rank, world)
    block = be.tensor(full[rank * blk:(rank + 1) * blk])
    for _ in range(3):   # repeated: buffers are recycled between calls
        proof, top = fp.prove(block)
    if rank == 0:
        want, wtop = eng.fri_prove(eng.fri_cfg(eng.prim_nth_root(N), s.G2, N, 8, 32), full.astype(np.uint64))
        ok = bytes(want) == proof and list(wtop) == top
        print("sharded prove", "OK" if ok else "MISMATCH", len(proof), "bytes", {k: round(v, 2) for k, v in fp.stage_ms.items()}, flush=True)
    # the whole build-defined prove (replicated LDE, sharded column trees + FRI) against one GPU's
    from stark_rs_amd.sharded import ShardedStarkProve
    logn, lb, W, t = logN - 3, 3, 4, 32
    cols = np.concatenate([(splitmix64(0x5354524B00 + c, 1 << logn) % np.uint64(p)).astype(np.uint32) for c in range(W)])
    trace = be.tensor(cols)
    sp = ShardedStarkProve(be, p, s.G2, logn, lb, W, t, eng.prim_nth_root(N), rank, world)
    for _ in range(2):
        roots, proof, top = sp.prove(trace)
    if rank == 0:
        want = eng.dev_stark_prove(trace.data_ptr(), W, logn, lb, t)
        ok = roots == [bytes(r) for r in want["column_roots"]] and proof == want["proof"] and top == want["top_indices"]
        print("sharded stark prove", "OK" if ok else "MISMATCH", len(proof), "bytes", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
