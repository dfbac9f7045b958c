// mgpu_loop.h -- the multi-GPU prover's round loop, written once against two small interfaces:
//   MgDev  -- the local device work (HIP kernels in the product, mgpu.hip; the CPU emulator in the
//             non-GPU tests, emu_mgpu.cpp)
//   MgColl -- the three collectives the path needs (RCCL over xGMI in the product; a caller-supplied
//             host shim for tests and other transports)
// One process per GPU; every rank runs the same calls in the same order.
//
// Fri::commit / Fri::prove over ONE codeword sharded in contiguous blocks (reference
// src/fri.rs:105-156, 250-311):
//   * tree: each rank builds the subtree over its block; the G sub-roots are all-gathered
//     (G x 32 bytes) and the log2 G levels above them are built on every rank;
//   * Fiat-Shamir: replicated on the device, no traffic and no host round trip;
//   * fold: out[i] needs c[i] and c[i + L/2] (src/fri.rs:80-84): output block g takes half a block
//     from rank g/2 and half a block from rank g/2 + G/2 -- one grouped send/recv per round in
//     which every rank sends each half of its block to one peer;
//   * once a block would drop below min_block elements the codeword is all-gathered and the
//     remaining rounds run replicated (identical results, no further traffic);
//   * query phase: openings are written by the rank that owns the leaf into a zeroed proof
//     buffer, one byte-sum all-reduce assembles the serialized proof (mgpu_core.h).
// The build-defined trace -> proof composition of stark.hip on top of it (mg_stark_prove), with
// the low-degree extension sharded by (column, coset) units:
//     E_c[r + B q] = NTT_n(coef_c[j] * Omega^(r j))[q]              (B = blowup, Omega^B = w_n)
// so a unit is an ordinary n-point coset transform; the units are dealt to the ranks, each plane
// is cut in G ranges of q and range g' travels to rank g' (one all-to-all), where interleaving
// the B cosets gives the natural-order block [g' N/G, (g'+1) N/G) of every column.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/stark_mi.h"
#include "mgpu_core.h"
#include "ntt_host.h"

struct MgXfer {
    int peer;
    void *ptr;      // device memory
    size_t bytes;
};

struct MgColl {
    virtual ~MgColl() {}
    // true: collectives are enqueued on the device's stream (RCCL); false: the loop drains the
    // stream before each call and the call has completed on return
    virtual bool stream_ordered() const = 0;
    virtual int all_gather(const void *send, void *recv, size_t bytes_per_rank) = 0;
    // grouped point-to-point: the k-th send to a peer matches that peer's k-th recv from this rank
    virtual int exchange(const std::vector<MgXfer> &sends, const std::vector<MgXfer> &recvs) = 0;
    virtual int all_reduce_sum_u8(void *buf, size_t bytes) = 0;
};

// One round of the fused tail (MgDev::fri_tail): tree over cw (len elements) into nodes, the Fiat-Shamir round of its
// root (MerkleRoot record at proof_slot, challenge to alpha_out unless null = last round), fold into next.
struct MgTailRound {
    const uint32_t *cw;
    uint32_t *next;        // nullptr on the last round
    uint8_t *nodes;
    uint8_t *proof_slot;
    uint64_t *alpha_out;   // nullptr on the last round
    uint64_t len, offset, omega;
};

struct MgDev {
    virtual ~MgDev() {}
    virtual uint32_t prime() const = 0;
    virtual uint32_t root_of_unity(uint32_t log_n) const = 0;       // forward primitive 2^log_n-th root
    virtual int fail(int code, const char *msg) = 0;
    // memory of one prove: bump-allocated, recycled by the next reset()
    virtual int reset() = 0;
    virtual void *alloc(size_t bytes) = 0;
    virtual int copy(void *dst, const void *src, size_t bytes) = 0;
    virtual int copy_rows(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, size_t rows) = 0;
    virtual int zero(void *dst, size_t bytes) = 0;
    virtual int upload(void *dst, const void *host, size_t bytes) = 0;
    virtual int download(void *host, const void *src, size_t bytes) = 0;   // synchronises
    virtual int sync() = 0;
    // Merkle (MerkleTree::new over leaf digests Hash::from_field_elements(&[v]), src/fri.rs:118-127)
    virtual int merkle(const uint32_t *elems, size_t n, uint8_t *nodes) = 0;
    virtual int merkle_fs(const uint32_t *elems, size_t n, uint8_t *nodes, void *fs, uint8_t *proof_slot, uint64_t *alpha_out) = 0;
    virtual int merkle_batch(const uint32_t *elems, size_t n, uint8_t *nodes, uint32_t n_trees, size_t elem_stride,
                             size_t node_stride_bytes) = 0;
    virtual int merkle_from_digests(size_t n, uint8_t *nodes) = 0;
    // FiatShamir (src/fiat_shamir.rs:15-25), state on the device
    virtual size_t fs_bytes() const = 0;
    virtual int fs_init(void *fs) = 0;
    virtual int fs_round(void *fs, const uint8_t *root, uint8_t *proof_slot, uint64_t *alpha_out) = 0;
    virtual int fs_challenge(const void *fs, uint64_t *out) = 0;
    virtual int fs_weights(const uint8_t *const *d_root_ptrs, uint32_t n, uint64_t *weights, uint8_t *roots_out) = 0;
    // FRI
    virtual int fold_shard(const uint32_t *lo, const uint32_t *hi, size_t count, size_t i0, size_t full_len, const uint64_t *alpha,
                           uint64_t offset, uint64_t omega, uint32_t *out) = 0;
    // The tail of Fri::commit (src/fri.rs:116-148) once the (replicated) codeword has at most tail_max_len() elements
    // and at most tail_max_rounds() rounds are left: every remaining round in one call.  The product launches the
    // single-workgroup fri_tail_kernel (csrc/hash.hip); the default is the per-round sequence, what the CPU
    // instantiation runs.  tail_max_len() == 0: never.
    virtual uint64_t tail_max_len() const { return 0; }
    virtual uint32_t tail_max_rounds() const { return 0; }
    virtual int fri_tail(const MgTailRound *rounds, uint32_t n_rounds, void *fs) {
        for (uint32_t k = 0; k < n_rounds; k++) {
            const MgTailRound &t = rounds[k];
            int rc = merkle_fs(t.cw, (size_t)t.len, t.nodes, fs, t.proof_slot, t.alpha_out);
            if (rc != SMI_OK) return rc;
            if (!t.next) break;
            rc = fold_shard(t.cw, t.cw + t.len / 2, (size_t)(t.len / 2), 0, (size_t)t.len, t.alpha_out, t.offset, t.omega, t.next);
            if (rc != SMI_OK) return rc;
        }
        return SMI_OK;
    }
    virtual int emit_codeword(const uint32_t *cw, uint64_t len, uint8_t *dst) = 0;
    virtual int sample_indices(const uint64_t *challenge, uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t *indices,
                               uint64_t *reduced) = 0;
    virtual int query(const MgLayer *layers_host, uint32_t n_layers, const uint64_t *top, uint32_t t, int rank, uint8_t *proof) = 0;
    virtual int column_open(const MgSide *cols_host, uint32_t W, const uint64_t *top, uint32_t t, int rank, uint8_t *out) = 0;
    // extension: lde = all columns whole (interpolate on trace_offset <w_n>, evaluate on lde_offset <w_N>)
    virtual int lde(const uint32_t *trace, uint32_t n_cols, uint32_t log_n, uint32_t log_b, uint64_t trace_offset, uint64_t lde_offset,
                    uint32_t *out) = 0;
    virtual int ntt(const uint32_t *in, uint32_t *out, uint32_t log_n, size_t n_in, uint32_t batch, size_t in_stride, size_t out_stride,
                    int inverse, uint64_t offset, uint64_t post_scale) = 0;
    // one transform sharded over 2^log_g ranks on the pass pipeline (ntt_driver.h)
    virtual int ntt_shard_first(uint32_t *strip, uint32_t log_n, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset) = 0;
    virtual int ntt_shard_rest(uint32_t *rows, uint32_t *out, uint32_t log_n, uint32_t log_g, int inverse) = 0;
    // out[c][(q << log_b) + r] = in[(c << log_b) + r][q], q < nq, r < 2^log_b, c < n_cols
    virtual int interleave(const uint32_t *in, uint32_t *out, uint32_t n_cols, uint32_t log_b, size_t nq) = 0;
    virtual int combine(const uint32_t *cols, uint32_t n_cols, size_t len, size_t stride, const uint64_t *weights, uint32_t *out) = 0;
};

struct MgFriOut {
    std::vector<uint8_t> proof;     // serialized ProofStream (every rank)
    std::vector<uint64_t> top;      // top-level indices (do_query)
    std::vector<uint64_t> alphas;   // R - 1 unreduced challenges
    uint64_t rounds = 0, last_len = 0;
};

namespace mg {
inline uint32_t ilog2(uint64_t n) {
    uint32_t l = 0;
    while ((n >> l) > 1) l++;
    return l;
}
inline bool pow2(uint64_t n) { return n && !(n & (n - 1)); }
#define MG_TRY(call)                 \
    do {                             \
        int rc__ = (call);           \
        if (rc__ != SMI_OK) return rc__; \
    } while (0)
inline int pre(MgDev &d, MgColl &c) { return c.stream_ordered() ? SMI_OK : d.sync(); }
}  // namespace mg

// Fri::commit (+ the query phase of Fri::prove when do_query) over the block this rank holds.
inline int mg_fri_run(MgDev &d, MgColl &coll, int rank, int G, const smi_fri_cfg &cfg, const uint32_t *block, size_t block_len,
                      size_t min_block, bool do_query, MgFriOut &out) {
    using namespace mg;
    // asserts of Fri::new / Fri::prove (src/fri.rs:37-45, 256-260)
    if (!pow2(cfg.domain_length)) return d.fail(SMI_ERR_DOMAIN_NOT_POW2, nullptr);
    if (!pow2(cfg.expansion_factor)) return d.fail(SMI_ERR_EXPANSION_NOT_POW2, nullptr);
    if (cfg.expansion_factor < 4) return d.fail(SMI_ERR_EXPANSION_TOO_SMALL, nullptr);
    if (G < 1 || !pow2((uint64_t)G) || rank < 0 || rank >= G) return d.fail(SMI_ERR_BAD_ARG, "mgpu: world size must be a power of two");
    const uint64_t N = cfg.domain_length;
    if ((uint64_t)block_len * (uint64_t)G != N) return d.fail(SMI_ERR_CODEWORD_LEN, "initial codeword length does not match domain length");
    const uint32_t p = d.prime();
    if (cfg.omega >= p || cfg.offset >= p) return d.fail(SMI_ERR_NON_CANONICAL, "omega/offset must be < p");
    uint64_t R = 0;
    for (uint64_t len = N; len > cfg.expansion_factor && 4 * cfg.num_colinearity_tests < len; len /= 2) R++;   // src/fri.rs:93-103
    if (R == 0) return d.fail(SMI_ERR_NO_ROUNDS, "num_rounds() == 0: the reference's verify rejects such a proof");
    const uint64_t t = cfg.num_colinearity_tests, last_n = N >> (R - 1);
    if (do_query) {   // asserts of src/fri.rs:183-192
        if (t > 2 * last_n) return d.fail(SMI_ERR_SAMPLE_ENTROPY, nullptr);
        if (t > last_n) return d.fail(SMI_ERR_SAMPLE_TOO_MANY, nullptr);
    }
    if (min_block < 2) min_block = 2;

    // proof layout (src/fri.rs:129,151,229-243; tags src/stream.rs:39-60) -- the single-GPU one
    const size_t off_last = 33 * R, off_layers = off_last + 9 + 8 * last_n;
    std::vector<MgLayer> layers(R - 1);
    size_t off = off_layers;
    for (uint64_t i = 0; i + 1 < R; i++) {
        const uint32_t dep = ilog2(N >> i);
        layers[i].off_triples = off;
        off += 33 * t;
        layers[i].off_paths = off;
        off += t * (2 * (9 + 32ull * dep) + (9 + 32ull * (dep - 1)));
    }
    const size_t proof_len = do_query ? off : off_layers;

    uint8_t *fs = (uint8_t *)d.alloc(d.fs_bytes());
    uint64_t *d_alphas = (uint64_t *)d.alloc(8 * (R + 1));
    uint64_t *d_seed = (uint64_t *)d.alloc(8);
    uint64_t *d_top = (uint64_t *)d.alloc(8 * (t + 1));
    uint64_t *d_red = (uint64_t *)d.alloc(8 * (t + 1));
    uint8_t *d_dummy = (uint8_t *)d.alloc(64);      // where ranks other than 0 drop the replicated root records
    uint8_t *d_proof = (uint8_t *)d.alloc(proof_len);
    if (!fs || !d_alphas || !d_seed || !d_top || !d_red || !d_dummy || !d_proof) return d.fail(SMI_ERR_OOM, "mgpu: device memory");
    MG_TRY(d.zero(d_proof, proof_len));
    MG_TRY(d.fs_init(fs));

    std::vector<MgSide> sides;
    const uint32_t *cur = block;
    uint64_t cur_local = block_len, length = N;
    bool sharded = G > 1;
    uint32_t omega = (uint32_t)cfg.omega, offset = (uint32_t)cfg.offset;
    const uint32_t logG = ilog2((uint64_t)G);
    for (uint64_t r = 0; r < R; r++) {
        if (sharded && cur_local < min_block) {        // small blocks: gather once, finish replicated
            uint32_t *full = (uint32_t *)d.alloc(length * 4);
            if (!full) return d.fail(SMI_ERR_OOM, "mgpu: codeword");
            MG_TRY(pre(d, coll));
            MG_TRY(coll.all_gather(cur, full, cur_local * 4));
            cur = full;
            cur_local = length;
            sharded = false;
        }
        if (!sharded && length <= d.tail_max_len() && R - r <= d.tail_max_rounds()) {
            // replicated and small: every remaining round in one call (SURVEY 8e "gather to one GPU and run the fused
            // tail"; here every rank runs it, so nothing has to be broadcast afterwards)
            std::vector<MgTailRound> tr((size_t)(R - r));
            for (uint64_t k = r; k < R; k++) {
                MgTailRound &t = tr[(size_t)(k - r)];
                const bool last_k = k == R - 1;
                t.cw = cur; t.len = length; t.offset = offset; t.omega = omega;
                t.nodes = (uint8_t *)d.alloc((2 * length - 1) * 32);
                t.proof_slot = rank == 0 ? d_proof + 33 * k : d_dummy;
                t.alpha_out = last_k ? nullptr : d_alphas + k;
                t.next = last_k ? nullptr : (uint32_t *)d.alloc((length / 2) * 4);
                if (!t.nodes || (!last_k && !t.next)) return d.fail(SMI_ERR_OOM, "mgpu: tail buffers");
                MgSide s;
                s.cw = cur; s.nodes = t.nodes; s.top = nullptr; s.len = length; s.blk = length;
                s.depth_local = ilog2(length); s.depth_top = 0;
                sides.push_back(s);
                if (last_k) break;
                cur = t.next;
                length /= 2;
                omega = host_mulmod(omega, omega, p);      // src/fri.rs:146-147
                offset = host_mulmod(offset, offset, p);
            }
            cur_local = length;
            MG_TRY(d.fri_tail(tr.data(), (uint32_t)tr.size(), fs));
            break;
        }
        uint8_t *nodes = (uint8_t *)d.alloc((2 * cur_local - 1) * 32);
        if (!nodes) return d.fail(SMI_ERR_OOM, "mgpu: tree");
        MgSide s;
        s.cw = cur; s.nodes = nodes; s.top = nullptr; s.len = length; s.blk = cur_local;
        s.depth_local = ilog2(cur_local); s.depth_top = 0;
        const bool last = r == R - 1;
        uint8_t *slot = rank == 0 ? d_proof + 33 * r : d_dummy;      // MerkleRoot record (src/fri.rs:129)
        uint64_t *alpha_out = last ? nullptr : d_alphas + r;         // no challenge after the last root (src/fri.rs:133-135)
        if (!sharded) {
            MG_TRY(d.merkle_fs(cur, cur_local, nodes, fs, slot, alpha_out));
        } else {
            MG_TRY(d.merkle(cur, cur_local, nodes));
            uint8_t *top = (uint8_t *)d.alloc((2 * (size_t)G - 1) * 32);
            if (!top) return d.fail(SMI_ERR_OOM, "mgpu: top tree");
            MG_TRY(pre(d, coll));
            MG_TRY(coll.all_gather(nodes + (2 * cur_local - 2) * 32, top, 32));
            MG_TRY(d.merkle_from_digests((size_t)G, top));
            MG_TRY(d.fs_round(fs, top + (2 * (size_t)G - 2) * 32, slot, alpha_out));
            s.top = top;
            s.depth_top = logG;
        }
        sides.push_back(s);
        if (last) break;
        const uint64_t half = length / 2;
        uint32_t *next;
        if (sharded) {
            const uint64_t hb = cur_local / 2;             // half a block; also the new block length
            const int g = rank, src_lo = g / 2, src_hi = g / 2 + G / 2, d0 = 2 * (g % (G / 2));
            uint32_t *own = const_cast<uint32_t *>(cur);
            uint32_t *rlo = src_lo == g ? nullptr : (uint32_t *)d.alloc(hb * 4);
            uint32_t *rhi = src_hi == g ? nullptr : (uint32_t *)d.alloc(hb * 4);
            next = (uint32_t *)d.alloc(hb * 4);
            if ((src_lo != g && !rlo) || (src_hi != g && !rhi) || !next) return d.fail(SMI_ERR_OOM, "mgpu: fold buffers");
            std::vector<MgXfer> sends, recvs;
            if (d0 != g) sends.push_back(MgXfer{d0, own, hb * 4});                 // first half  -> rank 2 (g mod G/2)
            if (d0 + 1 != g) sends.push_back(MgXfer{d0 + 1, own + hb, hb * 4});    // second half -> the next rank
            if (src_lo != g) recvs.push_back(MgXfer{src_lo, rlo, hb * 4});
            if (src_hi != g) recvs.push_back(MgXfer{src_hi, rhi, hb * 4});
            MG_TRY(pre(d, coll));
            MG_TRY(coll.exchange(sends, recvs));
            // what a rank keeps of its own block: rank 0 its first half as lo, rank G-1 its second half as hi
            const uint32_t *lo = src_lo == g ? own + (g & 1) * hb : rlo;
            const uint32_t *hi = src_hi == g ? own + (g & 1) * hb : rhi;
            MG_TRY(d.fold_shard(lo, hi, hb, (size_t)g * hb, length, d_alphas + r, offset, omega, next));
            cur_local = hb;
        } else {
            next = (uint32_t *)d.alloc(half * 4);
            if (!next) return d.fail(SMI_ERR_OOM, "mgpu: codeword");
            MG_TRY(d.fold_shard(cur, cur + half, half, 0, length, d_alphas + r, offset, omega, next));
            cur_local = half;
        }
        cur = next;
        length = half;
        omega = host_mulmod(omega, omega, p);      // src/fri.rs:146-147
        offset = host_mulmod(offset, offset, p);
    }
    if (sharded) {   // the last codeword goes out in the clear (src/fri.rs:151): every rank needs all of it
        uint32_t *full = (uint32_t *)d.alloc(length * 4);
        if (!full) return d.fail(SMI_ERR_OOM, "mgpu: codeword");
        MG_TRY(pre(d, coll));
        MG_TRY(coll.all_gather(cur, full, cur_local * 4));
        cur = full;
    }
    if (rank == 0) MG_TRY(d.emit_codeword(cur, length, d_proof + off_last));

    if (do_query) {
        MG_TRY(d.fs_challenge(fs, d_seed));                                   // src/fri.rs:272
        const uint64_t sample_size = R > 1 ? N / 2 : N;                         // src/fri.rs:266-270
        MG_TRY(d.sample_indices(d_seed, sample_size, last_n, (uint32_t)t, d_top, d_red));
        if (R > 1 && t > 0) {
            for (uint64_t i = 0; i + 1 < R; i++) {
                layers[i].cur = sides[i];
                layers[i].next = sides[i + 1];
            }
            MG_TRY(d.query(layers.data(), (uint32_t)(R - 1), d_top, (uint32_t)t, rank, d_proof));
        }
    }
    if (G > 1) {
        MG_TRY(pre(d, coll));
        MG_TRY(coll.all_reduce_sum_u8(d_proof, proof_len));
    }
    out.proof.resize(proof_len);
    out.alphas.assign(R, 0);
    out.top.assign(t + 1, 0);
    if (R > 1) MG_TRY(d.download(out.alphas.data(), d_alphas, 8 * (R - 1)));
    if (do_query && t) MG_TRY(d.download(out.top.data(), d_top, 8 * t));
    MG_TRY(d.download(out.proof.data(), d_proof, proof_len));
    out.alphas.resize(R - 1);
    out.top.resize(do_query ? t : 0);
    out.rounds = R;
    out.last_len = last_n;
    return SMI_OK;
}

// This rank's natural-order block [rank N/G, (rank+1) N/G) of the extension of every column: column c
// starts at *blocks_out + c * *stride_out.  trace: all W columns of n residues, on every rank.
// Sharding by (column, coset) units moves (G-1)/G of the extended trace once over the links: per rank
// (G-1)/G^2 of W*N*4 bytes spread over G-1 links.  At G = 8 that is 8 MB per link (0.1 ms) for an
// eighth of the transform work; at G = 2 it would be 128 MB over ONE link (about 1.8 ms at 70 GB/s per
// direction) to save 0.4 ms of a 0.8 ms extension -- so up to 2 ranks every rank extends all columns
// itself (no exchange) and only the hashing is sharded; from 4 ranks on the extension is sharded too.
inline int mg_lde_blocks(MgDev &d, MgColl &coll, int rank, int G, const uint32_t *trace, uint32_t W, uint32_t log_n, uint32_t log_b,
                         uint64_t trace_offset, uint64_t lde_offset, uint32_t **blocks_out, size_t *stride_out, int shard_from = 4) {
    using namespace mg;
    const uint64_t n = 1ull << log_n, B = 1ull << log_b, U = (uint64_t)W * B;
    const uint32_t p = d.prime();
    if (G < shard_from || G == 1) {   // the ordinary batched extension; this rank's blocks are views into it
        uint32_t *ext = (uint32_t *)d.alloc((size_t)W * n * B * 4);
        if (!ext) return d.fail(SMI_ERR_OOM, "mgpu: extension");
        MG_TRY(d.lde(trace, W, log_n, log_b, trace_offset, lde_offset, ext));
        *blocks_out = ext + (size_t)rank * ((n * B) / (uint64_t)G);
        *stride_out = (size_t)(n * B);
        return SMI_OK;
    }
    *stride_out = (size_t)((n * B) / (uint64_t)G);
    if (U % (uint64_t)G || n % (uint64_t)G) return d.fail(SMI_ERR_BAD_ARG, "mgpu: columns x blowup and the trace length must be multiples of the world size");
    const uint64_t upg = U / G, u0 = (uint64_t)rank * upg, nq = n / G;
    const uint64_t c_lo = u0 / B, c_hi = (u0 + upg - 1) / B, nc = c_hi - c_lo + 1;
    // interpolate my columns; the coset shift of the evaluation domain rides in the output scale
    uint32_t *coef = (uint32_t *)d.alloc(nc * n * 4);
    uint32_t *planes = (uint32_t *)d.alloc(upg * n * 4);
    uint32_t *recv = (uint32_t *)d.alloc(U * nq * 4);
    uint32_t *blocks = (uint32_t *)d.alloc(U * nq * 4);
    if (!coef || !planes || !recv || !blocks) return d.fail(SMI_ERR_OOM, "mgpu: extension");
    MG_TRY(d.ntt(trace + c_lo * n, coef, log_n, n, (uint32_t)nc, n, n, 1, trace_offset, lde_offset));
    const uint32_t Omega = d.root_of_unity(log_n + log_b);
    for (uint64_t j = 0; j < upg; j++) {          // unit u = c * B + r: the n-point transform on the coset Omega^r <w_n>
        const uint64_t u = u0 + j, c = u / B, r = u % B;
        MG_TRY(d.ntt(coef + (c - c_lo) * n, planes + j * n, log_n, n, 1, n, n, 0, host_powmod(Omega, r, p), 1));
    }
    std::vector<MgXfer> sends, recvs;
    for (int peer = 0; peer < G; peer++) {
        if (peer == rank) continue;
        for (uint64_t j = 0; j < upg; j++) {
            sends.push_back(MgXfer{peer, planes + j * n + (uint64_t)peer * nq, nq * 4});
            recvs.push_back(MgXfer{peer, recv + ((uint64_t)peer * upg + j) * nq, nq * 4});
        }
    }
    MG_TRY(d.copy_rows(recv + u0 * nq, nq * 4, planes + (uint64_t)rank * nq, n * 4, nq * 4, upg));   // my own ranges
    MG_TRY(pre(d, coll));
    MG_TRY(coll.exchange(sends, recvs));
    MG_TRY(d.interleave(recv, blocks, W, log_b, nq));
    *blocks_out = blocks;
    return SMI_OK;
}

// ONE transform of 2^log_n points over the G ranks (SURVEY 8e "one large NTT", BASELINE configs[3]) on the
// ordinary pass pipeline: with R_0 the plan's first digit and B = N / R_0,
//   in : strip = this rank's columns [rank B/G, (rank+1) B/G) of the row-major [R_0][B] view of the input,
//        as [R_0][B/G] (clobbered);
//   1. pass 0 on the strip (its inter-pass twiddle w_N^(k_0 b) is the four-step twiddle);
//   2. ONE all-to-all: rows k_0 in [h R_0/G, (h+1) R_0/G) of every strip go to rank h (each rank sends
//      (G-1)/G of its N/G elements; point-to-point over xGMI), laid out as [R_0/G][B];
//   3. the remaining passes, which are those of an (N/G)-point transform with first digit R_0/G;
//   out: X[k_0 + R_0 * rest] at rest * (R_0/G) + (k_0 - rank R_0/G): natural-order runs of R_0/G outputs;
//   natural: one more all-to-all of the same volume turns those runs into this rank's contiguous natural-order block
//        X[rank N/G .. (rank+1) N/G) -- the layout smi_mgpu_fri_* and the sharded Merkle trees consume.  (With ONE
//        exchange the rank that owns k_0 holds every N/R_0-th output; owning a contiguous range of outputs means owning
//        a range of `rest`, which is the other axis -- hence the second exchange: chunk d of the local result, the
//        outputs with rest in [d B/G, (d+1) B/G), goes to rank d and lands as rows of R_0/G in rows of R_0.)
// At G = 1 this is the direct transform, launch for launch.
inline int mg_ntt(MgDev &d, MgColl &coll, int rank, int G, uint32_t *strip, uint32_t *out, uint32_t log_n, int inverse, uint64_t offset,
                  bool natural = false) {
    using namespace mg;
    if (G < 1 || !pow2((uint64_t)G) || rank < 0 || rank >= G) return d.fail(SMI_ERR_BAD_ARG, "mgpu: world size must be a power of two");
    const uint32_t log_g = ilog2((uint64_t)G);
    if (!ntt_shard_ok(log_n, log_g)) return d.fail(SMI_ERR_BAD_ARG, "mgpu: transform too small to shard over this many ranks");
    const NttPlan pl = ntt_make_plan(log_n, 1);
    const uint64_t n = 1ull << log_n, R0 = 1ull << pl.logr[0], B = n / R0;
    const uint64_t rows = R0 / G, cols = B / G, blk = rows * cols;     // the block one rank sends to one rank
    MG_TRY(d.ntt_shard_first(strip, log_n, log_g, (uint32_t)rank, inverse, offset));
    uint32_t *mine = strip;
    if (G > 1) {
        uint32_t *stage = (uint32_t *)d.alloc((n / G) * 4), *rowsbuf = (uint32_t *)d.alloc((n / G) * 4);
        if (!stage || !rowsbuf) return d.fail(SMI_ERR_OOM, "mgpu: transform buffers");
        std::vector<MgXfer> sends, recvs;
        for (int peer = 0; peer < G; peer++) {
            if (peer == rank) continue;
            sends.push_back(MgXfer{peer, strip + (uint64_t)peer * blk, blk * 4});
            recvs.push_back(MgXfer{peer, stage + (uint64_t)peer * blk, blk * 4});
        }
        MG_TRY(pre(d, coll));
        MG_TRY(coll.exchange(sends, recvs));
        for (int g = 0; g < G; g++)      // [g][k_0][b_l] -> [k_0][g][b_l]
            MG_TRY(d.copy_rows(rowsbuf + (uint64_t)g * cols, B * 4, (g == rank ? strip : stage) + (uint64_t)g * blk, cols * 4, cols * 4, rows));
        mine = rowsbuf;
    }
    if (!natural || G == 1) return d.ntt_shard_rest(mine, out, log_n, log_g, inverse);
    uint32_t *cyc = (uint32_t *)d.alloc((n / G) * 4), *stage2 = (uint32_t *)d.alloc((n / G) * 4);
    if (!cyc || !stage2) return d.fail(SMI_ERR_OOM, "mgpu: transform buffers");
    MG_TRY(d.ntt_shard_rest(mine, cyc, log_n, log_g, inverse));
    const uint64_t chunk = (n / G) / G;                // = (B/G) * (R_0/G): the outputs this rank holds of one rank's block
    std::vector<MgXfer> sends, recvs;
    for (int peer = 0; peer < G; peer++) {
        if (peer == rank) continue;
        sends.push_back(MgXfer{peer, cyc + (uint64_t)peer * chunk, chunk * 4});
        recvs.push_back(MgXfer{peer, stage2 + (uint64_t)peer * chunk, chunk * 4});
    }
    MG_TRY(pre(d, coll));
    MG_TRY(coll.exchange(sends, recvs));
    for (int h = 0; h < G; h++)      // from rank h: [rest_l][k_0 - h R_0/G]  ->  out[rest_l * R_0 + h R_0/G + ...]
        MG_TRY(d.copy_rows(out + (uint64_t)h * rows, R0 * 4, (h == rank ? cyc : stage2) + (uint64_t)h * chunk, rows * 4, rows * 4, cols));
    return SMI_OK;
}

struct MgStarkOut {
    std::vector<uint8_t> column_roots;   // W x 32
    MgFriOut fri;
};

// trace -> proof, the composition of csrc/stark.hip (smi_dev_stark_prove, column trees) over G ranks:
// same column roots and the same proof bytes.
inline int mg_stark_prove(MgDev &d, MgColl &coll, int rank, int G, const smi_stark_cfg &cfg, const uint32_t *trace, size_t min_block,
                          MgStarkOut &out) {
    using namespace mg;
    const uint32_t W = cfg.n_cols, logN = cfg.log_n + cfg.log_blowup;
    if (!W || W > 64) return d.fail(SMI_ERR_BAD_ARG, "stark_prove: 1..64 columns");
    if (cfg.row_leaves) return d.fail(SMI_ERR_BAD_ARG, "mgpu: the row-leaf variant is single-GPU only");
    if (cfg.log_blowup < 2) return d.fail(SMI_ERR_EXPANSION_TOO_SMALL, nullptr);    // Fri::new, src/fri.rs:45
    if (G < 1 || !pow2((uint64_t)G) || rank < 0 || rank >= G) return d.fail(SMI_ERR_BAD_ARG, "mgpu: world size must be a power of two");
    const uint64_t N = 1ull << logN, blk = N / G;
    if (!blk) return d.fail(SMI_ERR_BAD_ARG, "mgpu: more ranks than leaves");
    MG_TRY(d.reset());
    uint32_t *ext = nullptr;
    size_t col_stride = 0;
    MG_TRY(mg_lde_blocks(d, coll, rank, G, trace, W, cfg.log_n, cfg.log_blowup, cfg.trace_offset, cfg.lde_offset, &ext, &col_stride));
    // one tree per column over this rank's leaves (one element per leaf, src/fri.rs:118-121), one set of launches
    const size_t tree_stride = 2 * blk * 32;
    uint8_t *trees = (uint8_t *)d.alloc(tree_stride * W);
    const uint8_t **d_rootp = (const uint8_t **)d.alloc(sizeof(void *) * W);
    uint64_t *d_w = (uint64_t *)d.alloc(8 * W);
    uint8_t *d_roots = (uint8_t *)d.alloc(32 * (size_t)W);
    uint32_t *cw = (uint32_t *)d.alloc(blk * 4);
    if (!trees || !d_rootp || !d_w || !d_roots || !cw) return d.fail(SMI_ERR_OOM, "mgpu: device memory");
    MG_TRY(d.merkle_batch(ext, blk, trees, W, col_stride, tree_stride));
    std::vector<const uint8_t *> rootp(W);
    if (G == 1) {
        for (uint32_t c = 0; c < W; c++) rootp[c] = trees + c * tree_stride + (2 * blk - 2) * 32;
    } else {
        // G x W sub-roots in one all-gather, then the log2 G upper levels of every column tree
        const size_t top_stride = (2 * (size_t)G - 1) * 32;
        uint8_t *sub = (uint8_t *)d.alloc(32 * (size_t)W), *all = (uint8_t *)d.alloc(32 * (size_t)W * G);
        uint8_t *tops = (uint8_t *)d.alloc(top_stride * W);
        if (!sub || !all || !tops) return d.fail(SMI_ERR_OOM, "mgpu: device memory");
        MG_TRY(d.copy_rows(sub, 32, trees + (2 * blk - 2) * 32, tree_stride, 32, W));
        MG_TRY(pre(d, coll));
        MG_TRY(coll.all_gather(sub, all, 32 * (size_t)W));
        for (uint32_t c = 0; c < W; c++) {
            MG_TRY(d.copy_rows(tops + c * top_stride, 32, all + 32 * (size_t)c, 32 * (size_t)W, 32, (size_t)G));
            MG_TRY(d.merkle_from_digests((size_t)G, tops + c * top_stride));
            rootp[c] = tops + c * top_stride + (2 * (size_t)G - 2) * 32;
        }
    }
    // weight c = FiatShamir::challenge after absorbing roots[0..c] (fresh transcript, csrc/stark.hip)
    MG_TRY(d.upload(d_rootp, rootp.data(), sizeof(void *) * W));
    MG_TRY(d.fs_weights(d_rootp, W, d_w, d_roots));
    MG_TRY(d.combine(ext, W, blk, col_stride, d_w, cw));
    smi_fri_cfg fc;
    fc.omega = d.root_of_unity(logN);
    fc.offset = cfg.lde_offset;
    fc.domain_length = N;
    fc.expansion_factor = 1ull << cfg.log_blowup;
    fc.num_colinearity_tests = cfg.num_colinearity_tests;
    MG_TRY(mg_fri_run(d, coll, rank, G, fc, cw, blk, min_block, true, out.fri));
    if (cfg.open_columns && cfg.num_colinearity_tests) {   // bind the codeword to the columns (mgpu_core.h)
        const uint32_t t = (uint32_t)cfg.num_colinearity_tests;
        std::vector<MgSide> sides(W);
        for (uint32_t c = 0; c < W; c++) {
            MgSide &sd = sides[c];
            sd.cw = ext + (size_t)c * col_stride; sd.nodes = trees + c * tree_stride; sd.len = N; sd.blk = blk;
            sd.depth_local = ilog2(blk);
            sd.top = G == 1 ? nullptr : rootp[c] - (2 * (size_t)G - 2) * 32;
            sd.depth_top = G == 1 ? 0 : ilog2((uint64_t)G);
        }
        const size_t ob = (size_t)mg_column_open_bytes(W, t, logN);
        uint64_t *d_top = (uint64_t *)d.alloc(8 * (size_t)t);
        uint8_t *d_open = (uint8_t *)d.alloc(ob);
        if (!d_top || !d_open) return d.fail(SMI_ERR_OOM, "mgpu: column openings");
        MG_TRY(d.upload(d_top, out.fri.top.data(), 8 * (size_t)t));
        MG_TRY(d.zero(d_open, ob));
        MG_TRY(d.column_open(sides.data(), W, d_top, t, rank, d_open));
        if (G > 1) {
            MG_TRY(pre(d, coll));
            MG_TRY(coll.all_reduce_sum_u8(d_open, ob));
        }
        const size_t at = out.fri.proof.size();
        out.fri.proof.resize(at + ob);
        MG_TRY(d.download(out.fri.proof.data() + at, d_open, ob));
    }
    out.column_roots.resize(32 * (size_t)W);
    return d.download(out.column_roots.data(), d_roots, 32 * (size_t)W);
}
