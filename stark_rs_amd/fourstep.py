"""Four-step NTT of one transform sharded over the GPUs of a node (BASELINE cfg 4).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  N = R*C points are
viewed as an R x C row-major matrix x[r*C + c]:

    rank g owns columns [g*C/G, (g+1)*C/G), stored column-major          (input layout)
      1. length-R NTT of every local column                (local, smi_dev_ntt)
      2. times offset^c * w_N^(kr*c), packed per destination (local, smi_dev_fourstep_twiddle_pack)
      3. all-to-all: rank h receives rows kr in [h*R/G, (h+1)*R/G)    (the ONE exchange step)
      4. transpose to row-major, length-C NTT of every row, transpose back (local)
    rank h ends with the [C][R/G] matrix whose row kc is X[kc*R + h*R/G + i], i < R/G:
    natural-order runs of R/G outputs, i.e. whole Merkle subtrees stay on one GPU.

Per GPU the exchange moves (G-1)/G of its N/G * 4 bytes in each direction; on MI355X the 7 xGMI
links are point-to-point, so a direct all-to-all (not a ring) uses all of them at once.

The local steps go through a small backend so the same sequencing runs on CPU tensors with
gloo in the tests (tests/test_fourstep_gloo.py); the product backend is HipBackend.
"""
import torch
import torch.distributed as dist


class HipBackend:
    """Local steps on the GPU through the C ABI (device pointers of int32 torch tensors)."""

    def __init__(self, engine):
        self.eng = engine
        # share torch's stream: buffers pass between torch ops / RCCL and the engine's kernels
        with torch.cuda.device(f"cuda:{engine.device}"):
            engine.set_stream(torch.cuda.current_stream().cuda_stream)

    def empty(self, n):
        return torch.empty(n, dtype=torch.int32, device=f"cuda:{self.eng.device}")

    def ntt_lines(self, buf, log_n, batch, inverse, offset):
        self.eng.dev_ntt(buf.data_ptr(), buf.data_ptr(), log_n, batch=batch, in_stride=1 << log_n, out_stride=1 << log_n,
                         inverse=inverse, offset=offset)

    def twiddle_pack(self, cols, send, log_r, log_c, c0, n_local, n_ranks, inverse, offset):
        self.eng.dev_fourstep_twiddle_pack(cols.data_ptr(), send.data_ptr(), log_r, log_c, c0, n_local, n_ranks, inverse, offset)

    def transpose(self, src, dst, rows, cols):
        self.eng.dev_transpose(src.data_ptr(), dst.data_ptr(), rows, cols)

    def fence(self):
        # RCCL may run on its own internal stream: drain ours before handing buffers over
        self.eng.sync()


class FourStepNTT:
    def __init__(self, backend, log_r, log_c, p, rank=0, world=1, group=None):
        assert world & (world - 1) == 0 and world <= (1 << log_r) and world <= (1 << log_c)
        self.b, self.log_r, self.log_c, self.rank, self.world, self.group = backend, log_r, log_c, rank, world, group
        self.p = p
        self.R, self.C = 1 << log_r, 1 << log_c
        self.cols_local, self.rows_local = self.C // world, self.R // world
        n_local = self.R * self.cols_local
        self.send = backend.empty(n_local)
        self.recv = backend.empty(n_local)

    def forward(self, cols, offset=1, inverse=False):
        """cols: this rank's C/G columns, column-major (modified in place).  Returns the [C][R/G]
        tensor described in the module docstring (a view of an internal buffer)."""
        b, w = self.b, self.world
        # coset: x[r*C + c] * offset^(r*C + c) = (offset^C)^r * offset^c; the first factor is the
        # column transform's own coset shift, the second rides along with the twiddle.
        assert not (inverse and offset != 1), "inverse four-step: offset must be 1"
        col_offset = pow(offset, self.C, self.p)
        b.ntt_lines(cols, self.log_r, self.cols_local, inverse, col_offset)
        b.twiddle_pack(cols, self.send, self.log_r, self.log_c, self.rank * self.cols_local, self.cols_local, w, inverse, offset)
        if w > 1:
            b.fence()
            dist.all_to_all_single(self.recv, self.send, group=self.group)
            if self.recv.is_cuda:
                torch.cuda.current_stream().synchronize()
            rows_in = self.recv
        else:
            rows_in = self.send
        # rows_in is [C][R/G]; make rows contiguous, transform, and return to natural-order runs
        b.transpose(rows_in, cols, self.C, self.rows_local)          # cols now holds [R/G][C]
        b.ntt_lines(cols, self.log_c, self.rows_local, inverse, 1)
        out = self.send
        b.transpose(cols, out, self.rows_local, self.C)              # [C][R/G]
        return out
