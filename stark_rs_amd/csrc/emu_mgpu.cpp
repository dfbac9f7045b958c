// emu_mgpu.cpp -- the multi-GPU round loop (mgpu_loop.h) on the CPU (TEST INFRASTRUCTURE).
//
// The build container has no GPU and the GPU box has one, so the loop that the product runs over
// HIP + RCCL (mgpu.hip) is also instantiated here over host memory: the device work is done with
// the same per-thread kernel code the emulator already drives (ntt_core.h, hash_core.h,
// fri_core.h, mgpu_core.h), the collectives go through the caller's functions (gloo in
// tests/test_mgpu_gloo.py).  What this checks is the part that cannot be exercised on one GPU: who
// sends what to whom, which rank writes which proof bytes, and that world sizes 2 and 4 give the
// oracle's bytes.  Part of libstarkmi_emu.so; the product never loads it.
#include <stdlib.h>

#include <string>
#include <vector>

#include "fri_core.h"
#include "hash_core.h"
#include "mgpu_loop.h"
#include "tables.h"

extern "C" {
int emu_ntt(uint64_t p, uint64_t g, const uint32_t *in, uint32_t *out, uint32_t L, uint32_t n_in, uint32_t batch, uint64_t in_stride,
            uint64_t out_stride, int inverse, uint64_t offset, uint64_t post_scale);
void emu_leaf_hash(const uint32_t *v, size_t n, uint8_t *out);
void emu_node_hash(const uint8_t *pairs, size_t n, uint8_t *out);
int emu_ntt_shard_first(uint64_t p, uint64_t g, uint32_t *strip, uint32_t L, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset);
int emu_ntt_shard_rest(uint64_t p, uint64_t g, uint32_t *rows, uint32_t *out, uint32_t L, uint32_t log_g, int inverse);
int emu_fold_shard(uint64_t p, uint64_t g, const uint32_t *lo, const uint32_t *hi, uint32_t count, uint32_t i0, uint32_t full_len,
                   uint64_t alpha, uint64_t offset, uint64_t omega, uint32_t *out);
}

namespace {
struct EmuDev : MgDev {
    uint64_t p, g;
    FieldSetup fs;
    std::vector<void *> blocks;
    std::string err;
    EmuDev(uint64_t p_, uint64_t g_) : p(p_), g(g_) { field_setup(p, g, &fs); }
    ~EmuDev() override { reset(); }
    uint32_t prime() const override { return (uint32_t)p; }
    uint32_t root_of_unity(uint32_t log_n) const override { return host_powmod(fs.wmax[0], 1ull << (fs.K - log_n), fs.F.p); }
    int fail(int code, const char *msg) override {
        err = msg ? msg : "";
        return code;
    }
    int reset() override {
        for (void *b : blocks) free(b);
        blocks.clear();
        return SMI_OK;
    }
    void *alloc(size_t bytes) override {
        void *q = calloc(bytes ? bytes : 4, 1);
        if (q) blocks.push_back(q);
        return q;
    }
    int copy(void *dst, const void *src, size_t bytes) override {
        memmove(dst, src, bytes);
        return SMI_OK;
    }
    int copy_rows(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, size_t rows) override {
        for (size_t r = 0; r < rows; r++) memmove((uint8_t *)dst + r * dst_pitch, (const uint8_t *)src + r * src_pitch, row_bytes);
        return SMI_OK;
    }
    int zero(void *dst, size_t bytes) override {
        memset(dst, 0, bytes);
        return SMI_OK;
    }
    int upload(void *dst, const void *host, size_t bytes) override { return copy(dst, host, bytes); }
    int download(void *host, const void *src, size_t bytes) override { return copy(host, src, bytes); }
    int sync() override { return SMI_OK; }

    int upper_levels(size_t n, uint8_t *nodes) {   // MerkleTree::new above level 0 (src/merkle.rs:21-31)
        size_t off = 0;
        for (size_t k = n; k > 1; k >>= 1) {
            emu_node_hash(nodes + off * 32, k / 2, nodes + (off + k) * 32);
            off += k;
        }
        return SMI_OK;
    }
    int merkle(const uint32_t *elems, size_t n, uint8_t *nodes) override {
        emu_leaf_hash(elems, n, nodes);
        return upper_levels(n, nodes);
    }
    int merkle_fs(const uint32_t *elems, size_t n, uint8_t *nodes, void *fsw, uint8_t *proof_slot, uint64_t *alpha_out) override {
        merkle(elems, n, nodes);
        return fs_round(fsw, nodes + (2 * n - 2) * 32, proof_slot, alpha_out);
    }
    int merkle_batch(const uint32_t *elems, size_t n, uint8_t *nodes, uint32_t n_trees, size_t elem_stride, size_t node_stride_bytes) override {
        for (uint32_t y = 0; y < n_trees; y++) merkle(elems + y * elem_stride, n, nodes + y * node_stride_bytes);
        return SMI_OK;
    }
    int merkle_from_digests(size_t n, uint8_t *nodes) override { return upper_levels(n, nodes); }

    size_t fs_bytes() const override { return 64; }
    int fs_init(void *fsw) override {
        hashc::State st;
        hashc::init(st);
        memcpy(fsw, st.s, 64);
        return SMI_OK;
    }
    int fs_round(void *fsw, const uint8_t *root, uint8_t *proof_slot, uint64_t *alpha_out) override {
        uint32_t m[8];
        memcpy(m, root, 32);
        hashc::fs_absorb_root((uint32_t *)fsw, m, proof_slot, alpha_out);
        return SMI_OK;
    }
    static uint64_t challenge_of(const hashc::State &at) {   // FiatShamir::challenge, src/fiat_shamir.rs:19-25
        hashc::State st = at;
        for (int k = 0; k < 8; k++) hashc::mix(st);
        uint32_t d[8];
        hashc::to_words(st, d);
        return (uint64_t)d[0] | ((uint64_t)d[1] << 32);
    }
    int fs_challenge(const void *fsw, uint64_t *out) override {
        hashc::State st;
        memcpy(st.s, fsw, 64);
        *out = challenge_of(st);
        return SMI_OK;
    }
    int fs_weights(const uint8_t *const *root_ptrs, uint32_t n, uint64_t *weights, uint8_t *roots_out) override {
        hashc::State st;
        hashc::init(st);
        for (uint32_t c = 0; c < n; c++) {
            uint32_t m[8];
            memcpy(m, root_ptrs[c], 32);
            memcpy(roots_out + 32 * c, m, 32);
            hashc::absorb_chunk32(st, m);
            weights[c] = challenge_of(st);
        }
        return SMI_OK;
    }
    int fold_shard(const uint32_t *lo, const uint32_t *hi, size_t count, size_t i0, size_t full_len, const uint64_t *alpha, uint64_t offset,
                   uint64_t omega, uint32_t *out) override {
        return emu_fold_shard(p, g, lo, hi, (uint32_t)count, (uint32_t)i0, (uint32_t)full_len, *alpha, offset, omega, out) ? SMI_ERR_BAD_ARG : SMI_OK;
    }
    // the loop's tail branch with the default per-round body (the product overrides fri_tail with one launch)
    uint64_t tail_max_len() const override { return 512; }
    uint32_t tail_max_rounds() const override { return 12; }
    int emit_codeword(const uint32_t *cw, uint64_t len, uint8_t *dst) override {   // src/fri.rs:151, src/stream.rs:48-53
        dst[0] = 2;
        mg_put_u64(dst + 1, len);
        for (uint64_t i = 0; i < len; i++) mg_put_u64(dst + 9 + 8 * i, cw[i]);
        return SMI_OK;
    }
    // Fri::sample_indices with seed = Hash::from_u64(challenge).0 (src/fri.rs:168-213, 272)
    int sample_indices(const uint64_t *challenge, uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t *indices,
                       uint64_t *reduced) override {
        uint8_t msg[36];
        for (int i = 0; i < 8; i++) msg[i] = (uint8_t)(*challenge >> (8 * i));
        uint32_t seed[8];
        hashc::hash_bytes(msg, 8, seed);
        memcpy(msg, seed, 32);
        uint32_t cnt = 0;
        for (uint32_t counter = 0; cnt < number; counter++) {
            for (int i = 0; i < 4; i++) msg[32 + i] = (uint8_t)(counter >> (8 * i));
            uint32_t d[8];
            hashc::hash_bytes(msg, 36, d);
            uint64_t acc = 0;   // sample_index: the last eight digest bytes, big-endian
            for (int i = 24; i < 32; i++) acc = (acc << 8) | ((d[i >> 2] >> (8 * (i & 3))) & 0xFFu);
            const uint64_t index = acc % size, ri = index % reduced_size;
            bool seen = false;
            for (uint32_t j = 0; j < cnt; j++) seen |= reduced[j] == ri;
            if (!seen) {
                indices[cnt] = index;
                reduced[cnt] = ri;
                cnt++;
            }
        }
        return SMI_OK;
    }
    int query(const MgLayer *layers, uint32_t n_layers, const uint64_t *top, uint32_t t, int rank, uint8_t *proof) override {
        for (uint32_t l = 0; l < n_layers; l++)
            for (uint32_t s = 0; s < t; s++) mg_query_write(layers[l], top[s], s, rank, proof, 0, 1);
        return SMI_OK;
    }
    int column_open(const MgSide *cols, uint32_t W, const uint64_t *top, uint32_t t, int rank, uint8_t *out) override {
        for (uint32_t s = 0; s < t; s++)
            for (uint32_t c = 0; c < W; c++) mg_column_open_write(cols, W, c, top[s], s, t, rank, out, 0, 1);
        return SMI_OK;
    }
    int lde(const uint32_t *trace, uint32_t n_cols, uint32_t log_n, uint32_t log_b, uint64_t trace_offset, uint64_t lde_offset,
            uint32_t *out) override {
        const uint64_t n = 1ull << log_n, N = n << log_b;
        if (ntt(trace, out, log_n, n, n_cols, n, N, 1, trace_offset, lde_offset)) return SMI_ERR_BAD_ARG;
        return ntt(out, out, log_n + log_b, n, n_cols, N, N, 0, 1, 1);
    }
    int ntt(const uint32_t *in, uint32_t *out, uint32_t log_n, size_t n_in, uint32_t batch, size_t in_stride, size_t out_stride, int inverse,
            uint64_t offset, uint64_t post_scale) override {
        std::vector<uint32_t> tmp;   // emu_ntt reads `in` while writing `out`: stage aliased input
        const uint32_t *src = in;
        if (in == out) {
            tmp.assign(in, in + (batch - 1) * in_stride + n_in);
            src = tmp.data();
        }
        return emu_ntt(p, g, src, out, log_n, (uint32_t)n_in, batch, in_stride, out_stride, inverse, offset, post_scale) ? SMI_ERR_BAD_ARG : SMI_OK;
    }
    int ntt_shard_first(uint32_t *strip, uint32_t log_n, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset) override {
        return emu_ntt_shard_first(p, g, strip, log_n, log_g, rank, inverse, offset) ? SMI_ERR_BAD_ARG : SMI_OK;
    }
    int ntt_shard_rest(uint32_t *rows, uint32_t *out, uint32_t log_n, uint32_t log_g, int inverse) override {
        return emu_ntt_shard_rest(p, g, rows, out, log_n, log_g, inverse) ? SMI_ERR_BAD_ARG : SMI_OK;
    }
    int interleave(const uint32_t *in, uint32_t *out, uint32_t n_cols, uint32_t log_b, size_t nq) override {
        const size_t B = (size_t)1 << log_b;
        for (size_t c = 0; c < n_cols; c++)
            for (size_t r = 0; r < B; r++)
                for (size_t q = 0; q < nq; q++) out[c * nq * B + q * B + r] = in[(c * B + r) * nq + q];
        return SMI_OK;
    }
    int combine(const uint32_t *cols, uint32_t n_cols, size_t len, size_t stride, const uint64_t *weights, uint32_t *out) override {
        const Fp &F = fs.F;
        for (size_t i = 0; i < len; i++) {
            uint32_t acc = 0;
            for (uint32_t c = 0; c < n_cols; c++)
                acc = fp_add(acc, mont_mul(cols[c * stride + i], to_mont((uint32_t)(weights[c] % F.p), F), F), F.p);
            out[i] = acc;
        }
        return SMI_OK;
    }
};

struct EmuColl : MgColl {
    smi_mgpu_coll ops;
    explicit EmuColl(const smi_mgpu_coll &o) : ops(o) {}
    bool stream_ordered() const override { return false; }
    int all_gather(const void *send, void *recv, size_t bytes_per_rank) override {
        return ops.all_gather(ops.user, send, recv, bytes_per_rank) ? SMI_ERR_RCCL : SMI_OK;
    }
    int exchange(const std::vector<MgXfer> &sends, const std::vector<MgXfer> &recvs) override {
        std::vector<int> sp, rp;
        std::vector<void *> sptr, rptr;
        std::vector<size_t> sb, rb;
        for (const MgXfer &x : sends) { sp.push_back(x.peer); sptr.push_back(x.ptr); sb.push_back(x.bytes); }
        for (const MgXfer &x : recvs) { rp.push_back(x.peer); rptr.push_back(x.ptr); rb.push_back(x.bytes); }
        return ops.exchange(ops.user, (int)sends.size(), sp.data(), sptr.data(), sb.data(), (int)recvs.size(), rp.data(), rptr.data(), rb.data())
                   ? SMI_ERR_RCCL
                   : SMI_OK;
    }
    int all_reduce_sum_u8(void *buf, size_t bytes) override { return ops.all_reduce_sum_u8(ops.user, buf, bytes) ? SMI_ERR_RCCL : SMI_OK; }
};

int put_proof(const std::vector<uint8_t> &bytes, uint8_t *proof, size_t cap, size_t *proof_len) {
    *proof_len = bytes.size();
    if (bytes.size() > cap) return SMI_ERR_BAD_ARG;
    memcpy(proof, bytes.data(), bytes.size());
    return SMI_OK;
}
}  // namespace

extern "C" int emu_mgpu_fri_prove(uint64_t p, uint64_t g, const smi_mgpu_coll *ops, int rank, int world, const smi_fri_cfg *cfg,
                                  const uint32_t *block, size_t block_len, size_t min_block, int do_query, uint8_t *proof, size_t cap,
                                  size_t *proof_len, uint64_t *top, uint64_t *alphas) {
    EmuDev d(p, g);
    EmuColl c(*ops);
    MgFriOut o;
    const int rc = mg_fri_run(d, c, rank, world, *cfg, block, block_len, min_block, do_query != 0, o);
    if (rc != SMI_OK) return rc;
    if (top && !o.top.empty()) memcpy(top, o.top.data(), 8 * o.top.size());
    if (alphas && !o.alphas.empty()) memcpy(alphas, o.alphas.data(), 8 * o.alphas.size());
    return put_proof(o.proof, proof, cap, proof_len);
}

extern "C" int emu_mgpu_stark_prove(uint64_t p, uint64_t g, const smi_mgpu_coll *ops, int rank, int world, const smi_stark_cfg *cfg,
                                    const uint32_t *trace, size_t min_block, uint8_t *column_roots, uint8_t *proof, size_t cap,
                                    size_t *proof_len, uint64_t *top) {
    EmuDev d(p, g);
    EmuColl c(*ops);
    MgStarkOut o;
    const int rc = mg_stark_prove(d, c, rank, world, *cfg, trace, min_block, o);
    if (rc != SMI_OK) return rc;
    if (column_roots) memcpy(column_roots, o.column_roots.data(), o.column_roots.size());
    if (top && !o.fri.top.empty()) memcpy(top, o.fri.top.data(), 8 * o.fri.top.size());
    return put_proof(o.fri.proof, proof, cap, proof_len);
}

extern "C" int emu_mgpu_lde(uint64_t p, uint64_t g, const smi_mgpu_coll *ops, int rank, int world, const uint32_t *trace, uint32_t n_cols,
                            uint32_t log_n, uint32_t log_b, uint64_t trace_offset, uint64_t lde_offset, uint32_t *out_blocks) {
    EmuDev d(p, g);
    EmuColl c(*ops);
    uint32_t *blocks = nullptr;
    size_t stride = 0;
    const int rc = mg_lde_blocks(d, c, rank, world, trace, n_cols, log_n, log_b, trace_offset, lde_offset, &blocks, &stride, 2);
    if (rc != SMI_OK) return rc;
    const size_t blk = ((size_t)1 << (log_n + log_b)) / (size_t)world;
    for (uint32_t col = 0; col < n_cols; col++) memcpy(out_blocks + col * blk, blocks + col * stride, blk * 4);
    return SMI_OK;
}

extern "C" int emu_mgpu_ntt(uint64_t p, uint64_t g, const smi_mgpu_coll *ops, int rank, int world, uint32_t *strip, uint32_t *out,
                            uint32_t log_n, int inverse, uint64_t offset) {
    EmuDev d(p, g);
    EmuColl c(*ops);
    return mg_ntt(d, c, rank, world, strip, out, log_n, inverse, offset);
}
extern "C" int emu_mgpu_ntt_natural(uint64_t p, uint64_t g, const smi_mgpu_coll *ops, int rank, int world, uint32_t *strip, uint32_t *out,
                                    uint32_t log_n, int inverse, uint64_t offset) {
    EmuDev d(p, g);
    EmuColl c(*ops);
    return mg_ntt(d, c, rank, world, strip, out, log_n, inverse, offset, true);
}
