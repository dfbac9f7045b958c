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

Fri::prove (src/fri.rs:250-311) on top of it (ShardedFriProve): index sampling is replicated
(every rank hashes the same transcript), each queried element and its authentication path come
from the rank that owns the leaf -- the path inside the local subtree from the resident nodes,
the top log2 G siblings from the all-gathered sub-roots -- and rank 0 serializes the proof in the
reference's object order (src/fri.rs:229-243, src/stream.rs:35-64): byte-identical to the
single-GPU proof.

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
        # One stream for everything: the codeword blocks are produced and consumed by torch ops
        # (cat, indexing, the collectives) and by the engine's kernels in turn; on two streams every
        # hand-over would need its own fence (a missed one after torch.cat let ranks diverge).
        with torch.cuda.device(self.dev):
            engine.set_stream(torch.cuda.current_stream().cuda_stream)

    def tensor(self, values):
        return torch.from_numpy(np.ascontiguousarray(values, dtype=np.uint32).view(np.int32)).to(self.dev)

    class Tree:
        """All levels of the Merkle tree over one device codeword block, resident on the device
        (level 0 first, MerkleTree.nodes of src/merkle.rs:18-33 flattened)."""

        def __init__(self, be, cw):
            self.n = cw.numel()
            self.nodes = torch.empty((2 * self.n - 1) * 32, dtype=torch.uint8, device=be.dev)
            be.eng.dev_merkle_build(cw.data_ptr(), self.n, self.nodes.data_ptr())
            be.eng.sync()
            self.root = bytes(self.nodes[-32:].cpu().numpy())

        def open_many(self, indices):
            """MerkleTree::open (src/merkle.rs:67-80) for a list of leaves: one device gather."""
            depth = self.n.bit_length() - 1
            if not indices or depth == 0:
                return [[] for _ in indices]
            rows = [[(2 * self.n - ((2 * self.n) >> l)) + ((i >> l) ^ 1) for l in range(depth)] for i in indices]
            idx = torch.tensor(rows, dtype=torch.int64, device=self.nodes.device)
            got = self.nodes.view(-1, 32)[idx.reshape(-1)].cpu().numpy().reshape(len(indices), depth, 32)
            return [[bytes(got[k, l]) for l in range(depth)] for k in range(len(indices))]

    def subtree(self, cw):
        return HipShardBackend.Tree(self, cw)

    def values(self, cw, indices):
        if not indices:
            return []
        idx = torch.tensor(indices, dtype=torch.int64, device=cw.device)
        return [int(v) & 0xFFFFFFFF for v in cw[idx].cpu().tolist()]

    def gather_many(self, reqs):
        """[(block, tree, local indices)] -> [(values, paths)], a path being depth*32 bytes
        (MerkleTree::open, src/merkle.rs:67-80): one index upload, one gather per tensor, one copy back."""
        rows, spans, at = [], [], 0
        for cw, tree, loc in reqs:
            depth = tree.n.bit_length() - 1
            spans.append((at, len(loc), depth))
            i = np.asarray(loc, dtype=np.int64)
            lv = np.arange(depth, dtype=np.int64)
            rows += [i, ((2 * tree.n - ((2 * tree.n) >> lv))[None, :] + ((i[:, None] >> lv[None, :]) ^ 1)).reshape(-1)]
            at += len(loc) * (1 + depth)
        if not at:
            return [([], []) for _ in reqs]
        idx = torch.from_numpy(np.concatenate(rows)).to(self.dev)
        pieces = []
        for (cw, tree, loc), (at, k, depth) in zip(reqs, spans):
            if not k:
                continue
            pieces.append(cw[idx[at:at + k]].view(torch.uint8))
            if depth:
                pieces.append(tree.nodes.view(-1, 32)[idx[at + k:at + k + k * depth]].reshape(-1))
        flat = torch.cat(pieces).cpu().numpy()
        out, at = [], 0
        for (_cw, _tree, _loc), (_a, k, depth) in zip(reqs, spans):
            vals = flat[at:at + 4 * k].view(np.uint32).tolist()
            at += 4 * k
            raw = flat[at:at + 32 * k * depth].tobytes()
            at += 32 * k * depth
            out.append((vals, [raw[32 * depth * j:32 * depth * (j + 1)] for j in range(k)]))
        return out

    def hash_pairs(self, digests):
        """[2k x 32] digests -> [k x 32]: Hash::combine of adjacent pairs on the device."""
        return self.eng.hash_combine_pairs(np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32))

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


def top_levels(backend, sub_roots):
    """Levels of the tree over the G sub-roots (level 0 = the sub-roots), replicated on every rank."""
    lv = [np.frombuffer(b"".join(sub_roots), dtype=np.uint8).reshape(-1, 32).copy()]
    while len(lv[-1]) > 1:
        lv.append(np.asarray(backend.hash_pairs(lv[-1]), dtype=np.uint8).reshape(-1, 32))
    return lv


def sample_index(digest, size):
    """src/fri.rs:168-174: a u128 accumulator shifted left by 8 per byte keeps, as usize, the last
    eight digest bytes big-endian."""
    return int.from_bytes(digest[-8:], "big") % size


def sample_indices(hash_bytes, seed, size, reduced_size, number):
    """src/fri.rs:176-213 (same asserts, same messages)."""
    assert number <= 2 * reduced_size, "not enough entropy in indices wrt last codeword"
    assert number <= reduced_size, "cannot sample more indices than available in last codeword"
    indices, reduced, counter = [], [], 0
    while len(indices) < number:
        index = sample_index(hash_bytes(seed + counter.to_bytes(4, "little")), size)
        counter += 1
        if index % reduced_size not in reduced:
            indices.append(index)
            reduced.append(index % reduced_size)
    return indices


class ShardedFriCommit:
    def __init__(self, backend, p, omega, offset, domain_length, expansion_factor, num_colinearity_tests, rank=0, world=1,
                 group=None, min_block=1 << 12):
        assert world & (world - 1) == 0 and domain_length % world == 0
        self.b, self.p, self.rank, self.world, self.group = backend, p, rank, world, group
        self.omega, self.offset, self.N = omega, offset, domain_length
        self.R = num_rounds(domain_length, expansion_factor, num_colinearity_tests)
        self.t = num_colinearity_tests
        self.min_block = max(min_block, 2)
        self.rounds = []     # with keep=True: per round {cw, tree, sharded, length, sub_roots}

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
    def commit(self, local_block, keep=False):
        """local_block: this rank's contiguous block of the initial codeword (int32 tensor of u32
        residues).  Returns (roots [R x bytes], alphas [R-1 unreduced ints], last codeword tensor).
        keep=True retains every round's local codeword block and tree for the query phase."""
        G, g, p = self.world, self.rank, self.p
        cw, length, sharded = local_block, self.N, G > 1
        omega, offset = self.omega, self.offset
        roots, alphas, transcript = [], [], b""
        self.rounds = []
        for r in range(self.R):
            if sharded and cw.numel() < self.min_block:
                cw, sharded = self._all_gather(cw), False
            tree = self.b.subtree(cw)
            subs = self._gather_roots(tree.root) if sharded else None
            tops = top_levels(self.b, subs) if sharded else None
            root = bytes(tops[-1][0]) if sharded else tree.root
            if keep:
                self.rounds.append({"cw": cw, "tree": tree, "sharded": sharded, "length": length, "sub_roots": subs, "top": tops})
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


class ShardedFriProve(ShardedFriCommit):
    """Fri::prove over the sharded commit.  Every rank calls prove(); rank 0 returns
    (serialized proof bytes, top-level indices), the other ranks (None, top-level indices)."""

    def prove(self, local_block):
        G, g, R, t = self.world, self.rank, self.R, self.t
        assert local_block.numel() * G == self.N, "initial codeword length does not match domain length"
        roots, _alphas, last = self.commit(local_block, keep=True)
        lens = [self.N >> i for i in range(R)]
        # replicated: challenge after the last root (src/fri.rs:272), seed = Hash::from_u64(challenge).0
        challenge = int.from_bytes(self.b.hash_bytes(b"".join(roots))[:8], "little")
        seed = self.b.hash_bytes(challenge.to_bytes(8, "little"))
        top = sample_indices(self.b.hash_bytes, seed, lens[1] if R > 1 else lens[0], lens[-1], t)

        # what this rank owns of every (layer, a/b/c, test) opening, grouped by the round whose block
        # and tree serve it (round r serves a and b of layer r and c of layer r-1)
        want, indices = [[] for _ in range(R)], list(top)
        for i in range(R - 1):
            half = lens[i] // 2
            indices = [x % half for x in indices]                                   # src/fri.rs:283-286
            for which, rnd, idxs in (("a", i, indices), ("b", i, [x + half for x in indices]), ("c", i + 1, indices)):
                rd = self.rounds[rnd]
                if rd["sharded"]:
                    blk = rd["length"] // G
                    want[rnd] += [((i, which, s_), j - g * blk) for s_, j in enumerate(idxs) if j // blk == g]
                elif g == 0:                             # replicated round: rank 0 has everything
                    want[rnd] += [((i, which, s_), j) for s_, j in enumerate(idxs)]
        reqs = [(rd["cw"], rd["tree"], [j for _, j in want[rnd]]) for rnd, rd in enumerate(self.rounds)]
        if hasattr(self.b, "gather_many"):               # one device -> host copy for the whole query phase
            res = self.b.gather_many(reqs)
        else:
            res = [(self.b.values(cw, loc), tree.open_many(loc)) for cw, tree, loc in reqs]
        mine = {}                                        # key -> (value, path as depth*32 bytes)
        for rnd, (vals, paths) in enumerate(res):
            rd = self.rounds[rnd]
            upper = b"".join(bytes(lv[(g >> l) ^ 1]) for l, lv in enumerate(rd["top"][:-1])) if rd["sharded"] else b""
            for (key, _), v, pth in zip(want[rnd], vals, paths):
                mine[key] = (v, (pth if isinstance(pth, (bytes, bytearray)) else b"".join(pth)) + upper)
        if G > 1:
            parts = [None] * G if g == 0 else None
            dist.gather_object(mine, parts, dst=0, group=self.group)
        else:
            parts = [mine]
        if g != 0:
            return None, top

        got = {}
        for part in parts:
            got.update(part)
        u64 = lambda v: int(v).to_bytes(8, "little")
        out = bytearray()
        for r_ in roots:                                                            # src/fri.rs:129
            out += b"\x00" + r_
        lastv = [int(v) & 0xFFFFFFFF for v in last.cpu().tolist()]
        out += b"\x02" + u64(len(lastv)) + b"".join(u64(v) for v in lastv)          # src/fri.rs:151
        for i in range(R - 1):
            for s_ in range(t):                                                     # src/fri.rs:229-236
                out += b"\x02" + u64(3) + b"".join(u64(got[(i, w, s_)][0]) for w in "abc")
            for s_ in range(t):                                                     # src/fri.rs:239-243
                for w in "abc":
                    pth = got[(i, w, s_)][1]
                    out += b"\x03" + u64(len(pth) // 32) + pth
        return bytes(out), top
