"""CPU, world_size 2 over gloo: the sequencing and buffer layouts of the sharded four-step NTT
(stark_rs_amd/fourstep.py) around a real all-to-all.  Local steps run through a CPU backend:
line transforms by the kernel emulator (the same phase code the HIP kernels run), twiddle/pack
and transposes restated in numpy.  The HIP versions of those two kernels are covered by the
-m gpu tests at world size 1."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

P2, G2 = 469762049, 3
u32p = C.POINTER(C.c_uint32)


class EmuBackend:
    def __init__(self, p, g):
        import stark_rs_amd as s
        s.build()
        self.L = C.CDLL(os.path.join(os.path.dirname(s.__file__), "build", "libstarkmi_emu.so"))
        self.L.emu_ntt.argtypes = [C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_uint64, C.c_int, C.c_uint64, C.c_uint64]
        self.p, self.g = p, g

    def empty(self, n):
        return torch.zeros(n, dtype=torch.int32)

    def _np(self, t):
        return t.numpy().view(np.uint32)

    def ntt_lines(self, buf, log_n, batch, inverse, offset):
        a = self._np(buf)
        out = np.zeros_like(a)
        rc = self.L.emu_ntt(self.p, self.g, a.ctypes.data_as(u32p), out.ctypes.data_as(u32p), log_n, 1 << log_n, batch,
                            1 << log_n, 1 << log_n, 1 if inverse else 0, offset, 1)
        assert rc == 0
        a[:] = out

    def twiddle_pack(self, cols, send, log_r, log_c, c0, n_local, n_ranks, inverse, offset):
        p, R = self.p, 1 << log_r
        N = 1 << (log_r + log_c)
        w = pow(self.g, (p - 1) // N, p)
        if inverse:
            w = pow(w, p - 2, p)
        a = self._np(cols).reshape(n_local, R).astype(object)
        rpg = R // n_ranks
        out = np.zeros((n_ranks, n_local, rpg), dtype=np.uint32)
        for cl in range(n_local):
            c = c0 + cl
            wc, oc = pow(w, c, p), pow(offset, c, p)
            f = oc
            for kr in range(R):
                out[kr // rpg, cl, kr % rpg] = int(a[cl, kr]) * f % p
                f = f * wc % p
        self._np(send)[:] = out.reshape(-1)

    def transpose(self, src, dst, rows, cols):
        self._np(dst)[:] = self._np(src).reshape(rows, cols).T.reshape(-1)

    def fence(self):
        pass


def _worker(rank, world, port, log_r, log_c, offset, inverse, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stark_rs_amd.fourstep import FourStepNTT
    from oracle import oracle as o
    R, Cc = 1 << log_r, 1 << log_c
    N = R * Cc
    x = (o.splitmix64(123, N) % np.uint64(P2)).astype(np.uint32)            # natural order x[r*C + c]
    ncl = Cc // world
    mine = x.reshape(R, Cc)[:, rank * ncl:(rank + 1) * ncl].T.copy()         # column-major local block
    fs = FourStepNTT(EmuBackend(P2, G2), log_r, log_c, P2, rank, world)
    out = fs.forward(torch.from_numpy(mine.reshape(-1).view(np.int32)), offset=offset, inverse=inverse)
    got = out.numpy().view(np.uint32).reshape(Cc, R // world).copy()         # row kc: X[kc*R + rank*R/G + i]
    gathered = [torch.zeros_like(torch.from_numpy(got.view(np.int32))) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(got.view(np.int32)))
    if rank == 0:
        full = np.zeros(N, dtype=np.uint64)
        rpg = R // world
        for h in range(world):
            blk = gathered[h].numpy().view(np.uint32)
            for kc in range(Cc):
                full[kc * R + h * rpg: kc * R + (h + 1) * rpg] = blk[kc]
        w = o.ff_prim_nth_root_g(N, P2, G2)
        if inverse:
            want = o.fast_intt(x.astype(np.uint64), w, 1, P2)
        else:
            want = o.fast_coset_ntt(x.astype(np.uint64), N, w, offset, P2)
        q.put(bool(np.array_equal(full, want)))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("log_r,log_c,offset,inverse", [(5, 4, 1, False), (4, 5, 31, False), (5, 5, 1, True)])
def test_four_step_world2_gloo(oracle, log_r, log_c, offset, inverse):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, log_r, log_c, offset, inverse, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True
