"""Fri::commit (reference src/fri.rs:105-156) for ONE codeword sharded over the GPUs of a node
(SURVEY 8e rows "Merkle tree" and "FRI fold").  One process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI).

Layout: the length-L codeword of round r is split in contiguous blocks, rank g holds
[g*L/G, (g+1)*L/G).  Per round:

  * Merkle: every rank builds the subtree over its own leaf block (device kernel), the G
    32-byte sub-roots are all-gathered (G*32 bytes -- the only tree traffic) and every rank
    combines them into the root with the same device hash (log2 G levels, replicated).  The
    digests are exactly those of MerkleTree::new over the whole codeword because a contiguous
    power-of-two leaf block is a complete subtree (src/merkle.rs:21-31).
  * Fiat-Shamir: every rank absorbs the same root and draws the same alpha (replicated, no
    traffic); the transcript hash runs on the device.
  * Fold: out[i] needs c[i] and c[i + L/2] (src/fri.rs:80-84).  Output block g needs half a
    block from rank g//2 and half a block from rank g//2 + G/2: one exchange in which every
    rank sends each half of its block to one peer (a perfect-shuffle permutation over the
    point-to-point xGMI links), then the shard fold kernel (smi_dev_fri_fold_shard).

When a block would drop below `min_block` elements the codeword is all-gathered once and the
remaining rounds run replicated on every rank (identical results, no further traffic).

Local work goes through a small backend so that the sequencing and the collectives are also
exercised on CPU tensors with gloo (tests/test_sharded_gloo.py); HipShardBackend is the product.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipShardBackend:
    """Local steps on the GPU through the C ABI."""

    def __init__(self, engine):
        self.eng = engine
        self.dev = f"cuda:{engine.device}"

    def tensor(self, values):
        return torch.from_numpy(np.ascontiguousarray(values, dtype=np.uint32).view(np.int32)).to(self.dev)

    def subtree_root(self, cw):
        """Merkle subtree over a device codeword block -> its 32-byte root (host bytes)."""
        n = cw.numel()
        nodes = torch.empty((2 * n - 1) * 32, dtype=torch.uint8, device=self.dev)
        self.eng.dev_merkle_build(cw.data_ptr(), n, nodes.data_ptr())
        self.eng.sync()
        return bytes(nodes[-32:].cpu().numpy())

    def combine_roots(self, roots):
        leaves = np.frombuffer(b"".join(roots), dtype=np.uint8).reshape(-1, 32)
        return self.eng.merkle_commit(leaves) if len(roots) > 1 else roots[0]

    def hash_bytes(self, data):
        return self.eng.hash_bytes(data)

    def fold(self, lo, hi, index0, full_len, alpha, offset, omega):
        out = torch.empty(lo.numel(), dtype=torch.int32, device=self.dev)
        a = torch.tensor([alpha - (1 << 64) if alpha >= (1 << 63) else alpha], dtype=torch.int64, device=self.dev)
        self.eng.dev_fri_fold_shard(lo.data_ptr(), hi.data_ptr(), lo.numel(), index0, full_len, a.data_ptr(), offset, omega,
                                    out.data_ptr())
        self.eng.sync()
        return out

    def fence(self):
        self.eng.sync()


def num_rounds(domain_length, expansion_factor, num_colinearity_tests):
    """src/fri.rs:93-103"""
    length, r = domain_length, 0
    while length > expansion_factor and 4 * num_colinearity_tests < length:
        length //= 2
        r += 1
    return r


class ShardedFriCommit:
    def __init__(self, backend, p, omega, offset, domain_length, expansion_factor, num_colinearity_tests, rank=0, world=1,
                 group=None, min_block=1 << 12):
        assert world & (world - 1) == 0 and domain_length % world == 0
        self.b, self.p, self.rank, self.world, self.group = backend, p, rank, world, group
        self.omega, self.offset, self.N = omega, offset, domain_length
        self.R = num_rounds(domain_length, expansion_factor, num_colinearity_tests)
        self.min_block = max(min_block, 2)

    # -- collectives ---------------------------------------------------------------------------
    def _gather_roots(self, root):
        if self.world == 1:
            return [root]
        out = [None] * self.world
        dist.all_gather_object(out, root, group=self.group)     # G x 32 bytes
        return out

    def _exchange_halves(self, block):
        """Perfect shuffle: rank s sends the first half of its block to rank 2*(s mod G/2) and the
        second half to the next rank; receives `lo` from rank g//2 and `hi` from g//2 + G/2."""
        G, g = self.world, self.rank
        half = block.numel() // 2
        recv = torch.empty_like(block)
        d0 = 2 * (g % (G // 2))
        in_split = [half if d in (d0, d0 + 1) else 0 for d in range(G)]
        out_split = [half if s in (g // 2, g // 2 + G // 2) else 0 for s in range(G)]
        self.b.fence()
        dist.all_to_all_single(recv, block, out_split, in_split, group=self.group)
        if recv.is_cuda:
            torch.cuda.current_stream().synchronize()
        return recv[:half], recv[half:]        # ordered by source rank: lo (g//2) then hi (g//2 + G/2)

    def _all_gather(self, block):
        parts = [torch.empty_like(block) for _ in range(self.world)]
        self.b.fence()
        dist.all_gather(parts, block, group=self.group)
        if block.is_cuda:
            torch.cuda.current_stream().synchronize()
        return torch.cat(parts)

    # -- the round loop ------------------------------------------------------------------------
    def commit(self, local_block):
        """local_block: this rank's contiguous block of the initial codeword (int32 tensor of u32
        residues).  Returns (roots [R x bytes], alphas [R-1 unreduced ints], last codeword tensor)."""
        G, g, p = self.world, self.rank, self.p
        cw, length, sharded = local_block, self.N, G > 1
        omega, offset = self.omega, self.offset
        roots, alphas, transcript = [], [], b""
        for r in range(self.R):
            if sharded and cw.numel() < self.min_block:
                cw, sharded = self._all_gather(cw), False
            sub = self.b.subtree_root(cw)
            root = self.b.combine_roots(self._gather_roots(sub)) if sharded else sub
            roots.append(root)
            transcript += root                                               # fiat_shamir.absorb, fri.rs:131
            if r == self.R - 1:
                break
            alpha = int.from_bytes(self.b.hash_bytes(transcript)[:8], "little")   # fiat_shamir.rs:19-25, unreduced
            alphas.append(alpha)
            half = length // 2
            if sharded:
                lo, hi = self._exchange_halves(cw)
                cw = self.b.fold(lo, hi, g * (half // G), length, alpha, offset, omega)
            else:
                cw = self.b.fold(cw[:half], cw[half:], 0, length, alpha, offset, omega)
            length = half
            omega, offset = omega * omega % p, offset * offset % p             # fri.rs:146-147
        if sharded:
            cw = self._all_gather(cw)
        return roots, alphas, cw
