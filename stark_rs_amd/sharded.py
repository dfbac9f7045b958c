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
            self.root_t = self.nodes[-32:]            # on the device: what the sub-root all-gather sends
            self._root = None

        @property
        def root(self):
            if self._root is None:
                self._root = bytes(self.root_t.cpu().numpy())
            return self._root

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

    class Transcript:
        """FiatShamir (src/fiat_shamir.rs:15-25) kept on the device: roots are copied in device to
        device, a challenge is the digest of the transcript so far written to device memory (its first
        8 bytes are the unreduced u64 the fold kernel reads) -- the round loop never waits for the host."""

        def __init__(self, be, max_roots):
            self.b, self.n, self.k = be, 0, 0
            self.buf = torch.empty(max(max_roots, 1) * 32, dtype=torch.uint8, device=be.dev)
            self.dig = torch.empty(max(max_roots, 1) * 32, dtype=torch.uint8, device=be.dev)

        def absorb(self, node):                       # node: a Tree or TopTree (root on the device)
            self.buf[32 * self.n:32 * self.n + 32].copy_(node.root_t)
            self.n += 1

        def challenge(self):                          # -> handle for fold(): 32 device bytes
            out = self.dig[32 * self.k:32 * self.k + 32]
            self.k += 1
            self.b.eng.dev_hash_bytes(self.buf.data_ptr(), 32 * self.n, out.data_ptr())
            return out

        def roots(self):
            flat = bytes(self.buf[:32 * self.n].cpu().numpy())
            return [flat[32 * i:32 * i + 32] for i in range(self.n)]

        def alphas(self):
            d = self.dig[:32 * self.k].cpu().numpy().reshape(-1, 32)
            return [int.from_bytes(bytes(r[:8]), "little") for r in d]

        def weights(self):                            # the challenges so far as k unreduced u64 on the device
            return self.dig[:32 * self.k].view(-1, 32)[:, :8].contiguous().view(torch.int64)

    def transcript(self, max_roots):
        return HipShardBackend.Transcript(self, max_roots)

    class TopTree:
        """The log2 G levels above the G sub-roots (replicated on every rank), built on the device."""

        def __init__(self, be, digests_t):
            self.G = digests_t.numel() // 32
            self.nodes = torch.empty((2 * self.G - 1) * 32, dtype=torch.uint8, device=be.dev)
            self.nodes[:self.G * 32].copy_(digests_t.reshape(-1))
            be.eng.dev_merkle_from_digests(self.G, self.nodes.data_ptr())
            self.root_t = self.nodes[-32:]

        def levels(self):
            flat, lv, off, k = self.nodes.cpu().numpy().reshape(-1, 32), [], 0, self.G
            while k >= 1:
                lv.append(flat[off:off + k])
                off, k = off + k, k // 2
            return lv

    def top_tree(self, digests_t):
        return HipShardBackend.TopTree(self, digests_t)

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

    def lde(self, trace, n_cols, log_n, log_blowup, trace_offset, lde_offset):
        """n_cols columns of 2^log_n residues (column-major) -> their extensions, column-major."""
        out = torch.empty(n_cols << (log_n + log_blowup), dtype=torch.int32, device=self.dev)
        self.eng.dev_lde(trace.data_ptr(), n_cols, log_n, log_blowup, out.data_ptr(), trace_offset, lde_offset)
        return out

    def combine(self, cols, n_cols, stride, start, length, weights):
        """sum_c (weights[c] mod p) * cols[c*stride + start + i], i < length (unreduced u64 weights)."""
        w = weights if isinstance(weights, torch.Tensor) else torch.tensor(
            [x - (1 << 64) if x >= (1 << 63) else x for x in weights], dtype=torch.int64, device=self.dev)
        out = torch.empty(length, dtype=torch.int32, device=self.dev)
        self.eng.dev_combine_columns(cols.data_ptr() + 4 * start, n_cols, length, stride, w.data_ptr(), out.data_ptr())
        return out

    def hash_pairs(self, digests):
        """[2k x 32] digests -> [k x 32]: Hash::combine of adjacent pairs on the device."""
        return self.eng.hash_combine_pairs(np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32))

    def hash_bytes(self, data):
        return self.eng.hash_bytes(data)

    def hash_bytes_batch(self, msgs):
        return self.eng.hash_bytes_batch(msgs)

    def fold(self, lo, hi, index0, full_len, alpha, offset, omega):
        out = torch.empty(lo.numel(), dtype=torch.int32, device=self.dev)
        a = alpha if isinstance(alpha, torch.Tensor) else torch.tensor(      # a Transcript challenge is already on the device
            [alpha - (1 << 64) if alpha >= (1 << 63) else alpha], dtype=torch.int64, device=self.dev)
        self.eng.dev_fri_fold_shard(lo.data_ptr(), hi.data_ptr(), lo.numel(), index0, full_len, a.data_ptr(), offset, omega,
                                    out.data_ptr())
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


class HostTranscript:
    """FiatShamir on the host, for backends without a device transcript (the CPU backend of the
    gloo tests): same interface as HipShardBackend.Transcript."""

    def __init__(self, backend):
        self.b, self._roots, self._alphas = backend, [], []

    def absorb(self, node):
        self._roots.append(node.root)                                         # fiat_shamir.absorb, fri.rs:131

    def challenge(self):
        a = int.from_bytes(self.b.hash_bytes(b"".join(self._roots))[:8], "little")   # fiat_shamir.rs:19-25, unreduced
        self._alphas.append(a)
        return a

    def roots(self):
        return list(self._roots)

    def alphas(self):
        return list(self._alphas)

    def weights(self):
        return list(self._alphas)


class HostTopTree:
    def __init__(self, backend, digests_t):
        d = bytes(digests_t.cpu().numpy())
        self._lv = top_levels(backend, [d[32 * i:32 * i + 32] for i in range(len(d) // 32)])
        self.root = bytes(self._lv[-1][0])

    def levels(self):
        return self._lv


def make_transcript(backend, max_roots):
    return backend.transcript(max_roots) if hasattr(backend, "transcript") else HostTranscript(backend)


def make_top_tree(backend, digests_t):
    return backend.top_tree(digests_t) if hasattr(backend, "top_tree") else HostTopTree(backend, digests_t)


def host_sync_needed(group=None):
    """RCCL collectives are ordered with the engine's kernels by the stream they share; any other
    backend (gloo in the tests and rehearsals) moves data on the host's schedule and needs the stream
    drained around each collective."""
    return dist.get_backend(group) != "nccl"


def gather_digests_t(backend, nodes, world, group=None):
    """All-gather of every rank's k subtree roots as one tensor collective -> uint8 tensor [G, k*32]
    on the backend's device; roots that are still on the device travel from there."""
    mine = torch.cat([n.root_t if hasattr(n, "root_t") else torch.frombuffer(bytearray(n.root), dtype=torch.uint8) for n in nodes])
    if world == 1:
        return mine.reshape(1, -1)
    parts = [torch.empty_like(mine) for _ in range(world)]
    if host_sync_needed(group):
        backend.fence()
    dist.all_gather(parts, mine, group=group)
    return torch.stack(parts)


def sample_index(digest, size):
    """src/fri.rs:168-174: a u128 accumulator shifted left by 8 per byte keeps, as usize, the last
    eight digest bytes big-endian."""
    return int.from_bytes(digest[-8:], "big") % size


def sample_indices(hash_bytes, seed, size, reduced_size, number, hash_batch=None):
    """src/fri.rs:176-213 (same asserts, same messages).  hash_batch (optional) digests a run of
    counters in one call; the accepted indices are the same either way."""
    assert number <= 2 * reduced_size, "not enough entropy in indices wrt last codeword"
    assert number <= reduced_size, "cannot sample more indices than available in last codeword"
    indices, reduced, counter, ready = [], set(), 0, []
    while len(indices) < number:
        if not ready:
            if hash_batch is None:
                ready = [hash_bytes(seed + counter.to_bytes(4, "little"))]
            else:
                run = max(2 * (number - len(indices)), 8)
                ready = hash_batch([seed + (counter + k).to_bytes(4, "little") for k in range(run)])
        index = sample_index(ready.pop(0), size)
        counter += 1
        if index % reduced_size not in reduced:
            indices.append(index)
            reduced.add(index % reduced_size)
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
        self.rounds = []     # with keep=True: per round {cw, tree, sharded, length, top}

    # -- collectives ---------------------------------------------------------------------------
    def _exchange_halves(self, block):
        """Perfect shuffle: rank s sends the first half of its block to rank 2*(s mod G/2) and the
        second half to the next rank; receives `lo` from rank g//2 and `hi` from g//2 + G/2."""
        G, g = self.world, self.rank
        half = block.numel() // 2
        recv = torch.empty_like(block)
        d0 = 2 * (g % (G // 2))
        in_split = [half if d in (d0, d0 + 1) else 0 for d in range(G)]
        out_split = [half if s in (g // 2, g // 2 + G // 2) else 0 for s in range(G)]
        sync = host_sync_needed(self.group)
        if sync:
            self.b.fence()
        dist.all_to_all_single(recv, block, out_split, in_split, group=self.group)
        if sync and recv.is_cuda:
            torch.cuda.current_stream().synchronize()
        return recv[:half], recv[half:]        # ordered by source rank: lo (g//2) then hi (g//2 + G/2)

    def _all_gather(self, block):
        parts = [torch.empty_like(block) for _ in range(self.world)]
        sync = host_sync_needed(self.group)
        if sync:
            self.b.fence()
        dist.all_gather(parts, block, group=self.group)
        if sync and block.is_cuda:
            torch.cuda.current_stream().synchronize()
        return torch.cat(parts)

    # -- the round loop ------------------------------------------------------------------------
    def commit(self, local_block, keep=False):
        """local_block: this rank's contiguous block of the initial codeword (int32 tensor of u32
        residues).  Returns (roots [R x bytes], alphas [R-1 unreduced ints], last codeword tensor).
        keep=True retains every round's local codeword block and tree for the query phase.
        With a device transcript (HIP backend) nothing in the loop waits for the host: roots and
        challenges are read back once, after the last round."""
        G, g, p = self.world, self.rank, self.p
        cw, length, sharded = local_block, self.N, G > 1
        omega, offset = self.omega, self.offset
        tr = make_transcript(self.b, self.R)
        self.rounds = []
        for r in range(self.R):
            if sharded and cw.numel() < self.min_block:
                cw, sharded = self._all_gather(cw), False
            tree = self.b.subtree(cw)
            # sharded: G sub-roots all-gathered (G x 32 bytes), the levels above them replicated
            top = make_top_tree(self.b, gather_digests_t(self.b, [tree], G, self.group).reshape(-1)) if sharded else None
            if keep:
                self.rounds.append({"cw": cw, "tree": tree, "sharded": sharded, "length": length, "top": top})
            tr.absorb(top if sharded else tree)                              # fiat_shamir.absorb, fri.rs:131
            if r == self.R - 1:
                break
            alpha = tr.challenge()                                           # fiat_shamir.rs:19-25, unreduced
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
        return tr.roots(), tr.alphas(), cw


class ShardedFriProve(ShardedFriCommit):
    """Fri::prove over the sharded commit.  Every rank calls prove(); rank 0 returns
    (serialized proof bytes, top-level indices), the other ranks (None, top-level indices)."""

    def prove(self, local_block):
        G, g, R, t = self.world, self.rank, self.R, self.t
        assert local_block.numel() * G == self.N, "initial codeword length does not match domain length"
        import time as _time
        marks = [("start", _time.perf_counter())]
        mark = lambda name: marks.append((name, _time.perf_counter()))
        roots, _alphas, last = self.commit(local_block, keep=True)
        mark("commit")
        lens = [self.N >> i for i in range(R)]
        # replicated: challenge after the last root (src/fri.rs:272), seed = Hash::from_u64(challenge).0
        challenge = int.from_bytes(self.b.hash_bytes(b"".join(roots))[:8], "little")
        seed = self.b.hash_bytes(challenge.to_bytes(8, "little"))
        top = sample_indices(self.b.hash_bytes, seed, lens[1] if R > 1 else lens[0], lens[-1], t,
                             getattr(self.b, "hash_bytes_batch", None))

        mark("sample")
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
        mark("gather")
        mine = {}                                        # key -> (value, path as depth*32 bytes)
        for rnd, (vals, paths) in enumerate(res):
            rd = self.rounds[rnd]
            upper = b"".join(bytes(lv[(g >> l) ^ 1]) for l, lv in enumerate(rd["top"].levels()[:-1])) if rd["sharded"] else b""
            for (key, _), v, pth in zip(want[rnd], vals, paths):
                mine[key] = (v, (pth if isinstance(pth, (bytes, bytearray)) else b"".join(pth)) + upper)
        mark("assemble")
        if G > 1:
            parts = [None] * G if g == 0 else None
            dist.gather_object(mine, parts, dst=0, group=self.group)
        else:
            parts = [mine]
        mark("collect")
        self.stage_ms = {b: 1e3 * (tb - ta) for (_, ta), (b, tb) in zip(marks, marks[1:])}
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
        self.stage_ms["serialize"] = 1e3 * (_time.perf_counter() - marks[-1][1])
        return bytes(out), top


class ShardedStarkProve:
    """The build-defined prove of csrc/stark.hip (LDE -> one tree per column -> Fiat-Shamir weights ->
    combined codeword -> Fri::prove) over G ranks, same column roots and proof bytes as one GPU.

    The hashing is 85 % of the single-GPU prove and the LDE 6 %, so the LDE is *replicated* (every rank
    extends all columns: no exchange, HBM-bound) and the hashing is *sharded*: rank g builds the
    subtree over leaves [g*N/G, (g+1)*N/G) of every column, the G x W sub-roots are all-gathered
    (G*W*32 bytes), the top log2 G levels and the transcript are replicated, every rank combines its own
    block of the columns, and ShardedFriProve takes it from there.  Every rank calls prove(); rank 0
    gets the proof bytes."""

    def __init__(self, backend, p, g, log_n, log_blowup, n_cols, num_colinearity_tests, omega_N, rank=0, world=1, group=None,
                 trace_offset=1, lde_offset=None, min_block=1 << 12):
        self.b, self.p, self.rank, self.world, self.group = backend, p, rank, world, group
        self.log_n, self.log_blowup, self.W = log_n, log_blowup, n_cols
        self.trace_offset, self.lde_offset = trace_offset, g if lde_offset is None else lde_offset
        self.N = 1 << (log_n + log_blowup)
        assert world & (world - 1) == 0 and self.N % world == 0
        self.fri = ShardedFriProve(backend, p, omega_N, self.lde_offset, self.N, 1 << log_blowup, num_colinearity_tests, rank, world,
                                   group, min_block)

    def prove(self, trace):
        """trace: the W columns of 2^log_n residues, column-major, the same on every rank.
        -> (column roots [W x bytes], proof bytes (rank 0) or None, top-level indices)."""
        G, g, W, N = self.world, self.rank, self.W, self.N
        blk = N // G
        lde = self.b.lde(trace, W, self.log_n, self.log_blowup, self.trace_offset, self.lde_offset)
        trees = [self.b.subtree(lde[c * N + g * blk:c * N + (g + 1) * blk]) for c in range(W)]
        subs = gather_digests_t(self.b, trees, G, self.group) if G > 1 else None     # G x W x 32 bytes
        # weight c = FiatShamir::challenge after absorbing roots[0..c] (csrc/stark.hip, fs_weights_kernel)
        tr = make_transcript(self.b, W)
        for c in range(W):
            tr.absorb(make_top_tree(self.b, subs[:, 32 * c:32 * c + 32].reshape(-1)) if G > 1 else trees[c])
            tr.challenge()
        block = self.b.combine(lde, W, N, g * blk, blk, tr.weights())
        roots = tr.roots()
        del trees                                                           # the column trees are not opened
        proof, top = self.fri.prove(block)
        return roots, proof, top
