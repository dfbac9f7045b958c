// fri.hip -- FRI fold kernel, on-device Fiat-Shamir / index sampling, query gather, and
// the host orchestration of Fri::commit / Fri::prove (reference src/fri.rs:57-311).
//
// The round loop of src/fri.rs:116-148 has a serial dependency root -> alpha -> fold
// (SURVEY H5).  Here the transcript state lives on the device: a single-lane kernel absorbs
// each root and draws alpha into device memory, the fold kernel reads alpha from there, so
// a whole prove is enqueued without one host round trip; the host synchronises once to
// copy the serialized proof back.
#include "fri_core.h"
#include "hash_core.h"
#include "hash_hex.h"
#include "internal.h"

int launch_merkle(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes);
// the same, with the Fiat-Shamir round of the root run by the launch that produces it (*done says whether it was)
int launch_merkle_fs(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t *fs_words, uint8_t *proof_slot,
                     uint64_t *alpha_out, bool *done);

// ------------------------------------------------------------------------- fold
// out[i] = 2^-1 * ((1 + a/x_i) c[i] + (1 - a/x_i) c[i+h])            (src/fri.rs:70-88)
//        = 2^-1 (c[i] + c[i+h]) + (a * 2^-1 * x_i^-1) (c[i] - c[i+h]),  x_i = offset * omega^i
// x_i^-1 comes from the two-level table S = offset^-1 * omega^-i; alpha is read from device
// memory (unreduced u64, src/fiat_shamir.rs:23-24) and reduced here.  HBM-bound: 12 B in,
// 4 B out per output element.
__global__ __launch_bounds__(256) void fri_fold_kernel(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi,
                                                       uint32_t *__restrict__ out, uint32_t count, uint32_t i0,
                                                       const uint64_t *__restrict__ alpha_ptr, Fp F, ScaleTables S, uint32_t inv2_m) {
    const uint32_t ah_m = fold_alpha_half(*alpha_ptr, inv2_m, F);
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += step)
        out[i] = fold_element(lo[i], hi[i], i0 + i, ah_m, inv2_m, S, F);
}

// ------------------------------------------------------------------------- Fiat-Shamir
// FiatShamir::{absorb,challenge} (src/fiat_shamir.rs:15-25).  Only whole 32-byte roots are
// ever absorbed (src/fri.rs:131), so the sponge state after k roots is carried in
// fs_state (16 words) and `challenge` = 8 more mixes on a copy: the same function of the
// whole transcript as re-hashing it, evaluated incrementally.
struct FsState {
    uint32_t s[16];
};

// (ride_*: a few bytes of the caller's -- the column roots of a prove -- placed behind the proof, to come back with its copy)
__global__ void fs_init_kernel(FsState *fs, const uint8_t *ride_src = nullptr, uint8_t *ride_dst = nullptr, size_t ride_n = 0) {
    if (blockIdx.x) return;
    for (size_t i = threadIdx.x; i < ride_n; i += blockDim.x) ride_dst[i] = ride_src[i];
    if (threadIdx.x) return;
    hashc::State st;
    hashc::init(st);
    for (int i = 0; i < 16; i++) fs->s[i] = st.s[i];
}

// absorb the root at `root`, append it (tag 0 + 32 bytes, src/stream.rs:39-42) to the proof
// buffer, and if alpha_out != nullptr draw the challenge.
// (both over the first sixteen lanes, one state word each: hash_hex.h)
__global__ void fs_round_kernel(FsState *fs, const uint32_t *root, uint8_t *proof_slot, uint64_t *alpha_out) {
    if (threadIdx.x >= 16 || blockIdx.x) return;
    const hashx::Lane row = hashx::make_lane(threadIdx.x);
    const uint32_t j = threadIdx.x >> 2;
    hashx::fs_absorb_root(fs->s, hashx::message(root[j], root[4 + j], row), row, proof_slot, alpha_out);
}

// challenge without absorbing (src/fri.rs:272: the index-sampling seed)
__global__ void fs_challenge_kernel(const FsState *fs, uint64_t *alpha_out) {
    if (threadIdx.x >= 16 || blockIdx.x) return;
    const hashx::Lane row = hashx::make_lane(threadIdx.x);
    const uint64_t a = hashx::low_bytes_u64(hashx::fs_challenge(fs->s[threadIdx.x], row));
    if (threadIdx.x == 0) *alpha_out = a;
}

// Fri::sample_indices (src/fri.rs:176-213) with seed = Hash::from_u64(challenge).0
// (src/fri.rs:272, src/hash.rs:37-39).  One wave: lane l hashes seed||counter for counter =
// 64*batch + l, then lane 0 accepts candidates in counter order exactly like the reference's
// sequential loop (distinct modulo reduced_size).  `number` <= reduced_size is checked on the host.
__global__ __launch_bounds__(64) void sample_indices_kernel(const uint64_t *challenge, uint64_t size, uint64_t reduced_size,
                                                            uint32_t number, uint64_t *indices, uint64_t *reduced) {
    __shared__ uint64_t cand[64], cand_r[64];
    __shared__ uint64_t s_red[256];   // accepted reduced indices, mirrored in LDS: the acceptance loop is O(number^2) look-ups
    __shared__ uint32_t s_cnt;
    const uint32_t lane = threadIdx.x;
    // seed = hash of the 8 LE bytes of the (unreduced) challenge; every lane computes it (uniform)
    const uint64_t ch = *challenge;
    hashc::State st;
    hashc::init(st);
#pragma unroll
    for (int i = 0; i < 8; i++) hashc::absorb_byte(st, i, (uint32_t)(ch >> (8 * i)) & 0xFFu);
    hashc::mix(st);
    for (int k = 0; k < 8; k++) hashc::mix(st);
    uint32_t seed[8];
    hashc::to_words(st, seed);
    hashc::State base;   // state after the first chunk (= seed) of every seed||counter message
    hashc::init(base);
    hashc::absorb_chunk32(base, seed);
    if (lane == 0) s_cnt = 0;
    uint64_t held[4] = {0, 0, 0, 0};
    __syncthreads();
    for (uint32_t batch = 0; batch < (1u << 20); batch++) {   // bounded: ends as soon as `number` are accepted
        const uint32_t counter = batch * 64u + lane;
        hashc::State s2 = base;
#pragma unroll
        for (int i = 0; i < 4; i++) hashc::absorb_byte(s2, i, (counter >> (8 * i)) & 0xFFu);
        hashc::mix(s2);
        for (int k = 0; k < 8; k++) hashc::mix(s2);
        uint32_t d[8];
        hashc::to_words(s2, d);
        // sample_index (src/fri.rs:168-174): u128 shift-xor over 32 bytes, truncated to usize
        // = the last 8 digest bytes read big-endian
        uint64_t acc = 0;
#pragma unroll
        for (int i = 24; i < 32; i++) acc = (acc << 8) | ((d[i >> 2] >> (8 * (i & 3))) & 0xFFu);
        // both moduli are powers of two on every FRI path (lengths of codewords): a mask instead of
        // a 64-bit division; every lane reduces its own candidate, lane 0 only compares
        const uint64_t index_l = (size & (size - 1)) ? acc % size : acc & (size - 1);
        cand[lane] = index_l;
        cand_r[lane] = (reduced_size & (reduced_size - 1)) ? index_l % reduced_size : index_l & (reduced_size - 1);
        __syncthreads();
        if (number <= 256) {
            // acceptance in candidate order, the look-up spread over the wave: lane l keeps the accepted
            // reduced indices l, l+64, l+128, l+192 in registers and a ballot answers "seen before?"
            uint32_t cnt = s_cnt;   // wave-uniform
            for (uint32_t k = 0; k < 64 && cnt < number; k++) {
                const uint64_t index = cand[k], ri = cand_r[k];
                bool hit = false;
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) hit |= lane + 64u * m < cnt && held[m] == ri;
                if (__ballot(hit) == 0) {
#pragma unroll
                    for (uint32_t m = 0; m < 4; m++)
                        if ((cnt >> 6) == m && (cnt & 63u) == lane) held[m] = ri;
                    if (lane == 0) {
                        indices[cnt] = index;
                        reduced[cnt] = ri;
                    }
                    cnt++;
                }
            }
            __syncthreads();
            if (lane == 0) s_cnt = cnt;
        } else if (lane == 0) {
            uint32_t cnt = s_cnt;
            for (uint32_t k = 0; k < 64 && cnt < number; k++) {
                const uint64_t index = cand[k], ri = cand_r[k];
                bool seen = false;
                for (uint32_t j = 0; j < cnt; j++) seen |= (j < 256 ? s_red[j] : reduced[j]) == ri;
                if (!seen) {
                    indices[cnt] = index;
                    reduced[cnt] = ri;
                    if (cnt < 256) s_red[cnt] = ri;
                    cnt++;
                }
            }
            s_cnt = cnt;
        }
        __syncthreads();
        if (s_cnt >= number) break;
    }
}

// ------------------------------------------------------------------------- query gather
// Fri::query (src/fri.rs:215-248) for every layer, written straight into the serialized
// ProofStream layout (src/stream.rs:35-64).  One workgroup per (layer, test).
struct LayerInfo {
    const uint32_t *cw, *cw_next;
    const uint8_t *nodes, *nodes_next;
    uint64_t len;           // of cw
    uint64_t off_triples;   // byte offset of this layer's first FieldElements triple
    uint64_t off_paths;     // byte offset of this layer's first MerklePath
    uint32_t depth, depth_next;
};

__device__ void put_u64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}
// MerkleTree::open (src/merkle.rs:67-80) serialized as tag 3, u64 count, 32 B digests
__device__ void write_path(uint8_t *dst, const uint8_t *nodes, uint64_t n, uint32_t depth, uint64_t index, uint32_t lane) {
    if (lane == 0) {
        dst[0] = 3;
        put_u64(dst + 1, depth);
    }
    uint64_t idx = index, lvl_off = 0, len = n;
    for (uint32_t l = 0; l < depth; l++) {
        const uint64_t sib = idx ^ 1;
        if (lane < 32) dst[9 + 32 * l + lane] = nodes[(lvl_off + sib) * 32 + lane];
        idx >>= 1;
        lvl_off += len;
        len >>= 1;
    }
}
// The layer table travels as a kernel argument while it fits (SMI_QUERY_TAB_MAX layers: codewords of up to 2^40
// elements): a host-to-device copy between the index sampling and this launch would sit on the critical path of
// the prove with the runtime's gaps around it (profiles/r03_b_prove_timeline.txt).
#define SMI_QUERY_TAB_MAX 40
struct LayerTable {
    LayerInfo l[SMI_QUERY_TAB_MAX];
};
__device__ __forceinline__ void query_one(const LayerInfo &L, const uint64_t *top, uint8_t *proof);
__global__ void query_kernel(const LayerInfo *layers, const uint64_t *top, uint32_t t, uint8_t *proof) {
    const LayerInfo L = layers[blockIdx.y];
    query_one(L, top, proof);
}
__global__ void query_tab_kernel(const LayerTable tab, const uint64_t *top, uint32_t t, uint8_t *proof) {
    const LayerInfo L = tab.l[blockIdx.y];
    query_one(L, top, proof);
}
__device__ __forceinline__ void query_one(const LayerInfo &L, const uint64_t *top, uint8_t *proof) {
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint64_t half = L.len / 2;
    const uint64_t c = top[s] % half;   // indices folded layer by layer: (x % a) % b == x % b for b | a
    if (lane == 0) {
        uint8_t *tr = proof + L.off_triples + (uint64_t)s * 33;
        tr[0] = 2;
        put_u64(tr + 1, 3);
        put_u64(tr + 9, L.cw[c]);
        put_u64(tr + 17, L.cw[c + half]);
        put_u64(tr + 25, L.cw_next[c]);
    }
    const uint64_t pa = 9 + 32ull * L.depth, pc = 9 + 32ull * L.depth_next;
    uint8_t *pp = proof + L.off_paths + (uint64_t)s * (2 * pa + pc);
    write_path(pp, L.nodes, L.len, L.depth, c, lane);
    write_path(pp + pa, L.nodes, L.len, L.depth, c + half, lane);
    write_path(pp + 2 * pa, L.nodes_next, half, L.depth_next, c, lane);
}

// last codeword in the clear: tag 2, u64 len, len x u64 (src/fri.rs:151, src/stream.rs:48-53)
__global__ void emit_codeword_kernel(const uint32_t *cw, uint64_t len, uint8_t *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        dst[0] = 2;
        put_u64(dst + 1, len);
    }
    if (i < len) put_u64(dst + 9 + 8 * i, cw[i]);
}

// launchers for the multi-GPU round loop (mgpu.hip), which sequences the same kernels
size_t fri_fs_bytes() { return sizeof(FsState); }
int launch_fs_init(smi_ctx *ctx, void *fs) {
    fs_init_kernel<<<1, 64, 0, ctx->stream>>>((FsState *)fs);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_fs_round(smi_ctx *ctx, void *fs, const uint8_t *root, uint8_t *proof_slot, uint64_t *alpha_out) {
    fs_round_kernel<<<1, 64, 0, ctx->stream>>>((FsState *)fs, (const uint32_t *)root, proof_slot, alpha_out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_fs_challenge(smi_ctx *ctx, const void *fs, uint64_t *out) {
    fs_challenge_kernel<<<1, 64, 0, ctx->stream>>>((const FsState *)fs, out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_sample_indices(smi_ctx *ctx, const uint64_t *challenge, uint64_t size, uint64_t reduced_size, uint32_t number,
                          uint64_t *indices, uint64_t *reduced) {
    sample_indices_kernel<<<1, 64, 0, ctx->stream>>>(challenge, size, reduced_size, number, indices, reduced);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_emit_codeword(smi_ctx *ctx, const uint32_t *cw, uint64_t len, uint8_t *dst) {
    emit_codeword_kernel<<<(uint32_t)((len + 255) / 256), 256, 0, ctx->stream>>>(cw, len, dst);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

// ------------------------------------------------------------------------- host side
int smi_fri_num_rounds(const smi_fri_cfg *cfg, uint64_t *rounds) {
    if (!cfg || !rounds) return SMI_ERR_BAD_ARG;
    uint64_t len = cfg->domain_length, r = 0;  // src/fri.rs:93-103
    while (len > cfg->expansion_factor && 4 * cfg->num_colinearity_tests < len) {
        len /= 2;
        r++;
    }
    *rounds = r;
    return SMI_OK;
}

static bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }
static uint32_t ilog2(uint64_t n) {
    uint32_t l = 0;
    while ((n >> l) > 1) l++;
    return l;
}

int smi_fri_check(const smi_ctx *ctx, const smi_fri_cfg *cfg) {
    if (!ctx || !cfg) return SMI_ERR_BAD_ARG;
    if (!is_pow2(cfg->domain_length)) return SMI_ERR_DOMAIN_NOT_POW2;      // src/fri.rs:37-40
    if (!is_pow2(cfg->expansion_factor)) return SMI_ERR_EXPANSION_NOT_POW2;  // src/fri.rs:41-44
    if (cfg->expansion_factor < 4) return SMI_ERR_EXPANSION_TOO_SMALL;       // src/fri.rs:45
    return SMI_OK;
}

struct smi_fri_run {
    smi_ctx *ctx;
    std::vector<uint32_t *> codewords;  // device, lengths N, N/2, ...
    std::vector<uint8_t *> trees;       // device nodes per codeword
    std::vector<uint64_t> lens;
    void *d_misc;                       // fs state, alphas, indices, layer table
    uint8_t *d_proof;
    bool owns_first;                    // codewords[0] allocated by us (vs caller's buffer)
    bool arena;                         // buffers live in the context arena (not retained past the call)
};
static void *run_alloc(smi_fri_run *run, size_t bytes) {
    if (run->arena) return arena_alloc(run->ctx, bytes);
    void *q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 4) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return q;
}

void smi_fri_run_free(smi_fri_run *run) {
    if (!run) return;
    DeviceGuard dg__(run->ctx);
    if (run->arena) {  // arena memory is recycled by the next arena_reset
        if (run->owns_first && !run->codewords.empty()) {
            (void)hipStreamSynchronize(run->ctx->stream);
            (void)hipFree(run->codewords[0]);
        }
        delete run;
        return;
    }
    (void)hipStreamSynchronize(run->ctx->stream);
    for (size_t i = 0; i < run->codewords.size(); i++)
        if (i > 0 || run->owns_first) (void)hipFree(run->codewords[i]);
    for (uint8_t *t : run->trees) (void)hipFree(t);
    (void)hipFree(run->d_misc);
    (void)hipFree(run->d_proof);
    delete run;
}

// Folds `count` outputs starting at global index i0 of a codeword of length full_len:
// out[k] = fold(lo[k], hi[k]) where lo[k] = c[i0+k], hi[k] = c[i0+k+full_len/2].
int launch_fold_shard(smi_ctx *ctx, const uint32_t *d_lo, const uint32_t *d_hi, size_t count, size_t i0, size_t full_len,
                             const uint64_t *d_alpha, uint64_t offset, uint64_t omega, uint32_t *d_out) {
    const uint32_t p = ctx->fs.F.p;
    if (full_len < 2 || !is_pow2(full_len)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "fold: codeword length must be a power of two >= 2");
    if (i0 + count > full_len / 2) return smi_fail(ctx, SMI_ERR_BAD_ARG, "fold: shard outside the folded codeword");
    if (offset >= p || omega >= p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "fold: offset/omega must be < p");
    if (offset == 0 || omega == 0) return smi_fail(ctx, SMI_ERR_DIV_BY_ZERO, "no division by zero");  // src/ff.rs:182
    if (!count) return SMI_OK;
    ScaleScope pin__(ctx);
    ScaleTables S;
    SMI_TRY(ctx_scale_tables(ctx, h_inv(ctx, (uint32_t)offset), h_inv(ctx, (uint32_t)omega), ilog2(full_len / 2), &S));
    const uint32_t inv2_m = (uint32_t)(((uint64_t)h_inv(ctx, 2) << 32) % p);
    uint32_t grid = (uint32_t)((count + 255) / 256);
    if (grid > 2048) grid = 2048;
    ProfScope ps(ctx, "fri_fold_kernel", 12.0 * (double)count);  // read 2 x 4 B, write 4 B per output
    fri_fold_kernel<<<grid, 256, 0, ctx->stream>>>(d_lo, d_hi, d_out, (uint32_t)count, (uint32_t)i0, d_alpha, ctx->fs.F, S, inv2_m);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_fold(smi_ctx *ctx, const uint32_t *d_in, size_t len, const uint64_t *d_alpha, uint64_t offset, uint64_t omega,
                uint32_t *d_out) {
    if (len < 2 || !is_pow2(len)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "fold: codeword length must be a power of two >= 2");
    return launch_fold_shard(ctx, d_in, d_in + len / 2, len / 2, 0, len, d_alpha, offset, omega, d_out);
}
int smi_dev_fri_fold_shard(smi_ctx *ctx, const uint32_t *d_lo, const uint32_t *d_hi, size_t count, size_t index0, size_t full_len,
                           const uint64_t *d_alpha, uint64_t offset, uint64_t omega, uint32_t *d_out) {
    if (!ctx || !d_lo || !d_hi || !d_alpha || !d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return launch_fold_shard(ctx, d_lo, d_hi, count, index0, full_len, d_alpha, offset, omega, d_out);
}

// misc device block layout (the challenges and the sampled indices are not here: they sit behind the proof, fri_run)
struct MiscLayout {
    size_t fs, seed_ch, reduced, layers, total;
};
static MiscLayout misc_layout(uint64_t R, uint64_t t) {
    MiscLayout m;
    size_t o = 0;
    m.fs = o; o += sizeof(FsState);
    o = (o + 63) & ~(size_t)63;
    m.seed_ch = o; o += 8;
    m.reduced = o; o += 8 * (t + 1);
    o = (o + 63) & ~(size_t)63;
    m.layers = o; o += sizeof(LayerInfo) * (R + 1);
    m.total = o;
    return m;
}

// Fri::commit (+ optionally the query phase of Fri::prove) over a device codeword.
// With do_query == false only roots/alphas/last codeword are produced.
uint64_t fri_tail_len() {
    static const uint64_t tail_len = [] {
        const char *e = getenv("SMI_FRI_TAIL");
        const uint64_t v = e ? (uint64_t)atoll(e) : 512;
        return v > SMI_FRI_TAIL_MAX_LEN ? (uint64_t)SMI_FRI_TAIL_MAX_LEN : v;
    }();
    return tail_len;
}
int fri_run(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint32_t *d_codeword, size_t len, bool do_query, bool reset_arena,
                   smi_fri_run **run_out, std::vector<uint8_t> *proof_host, uint64_t *top_host, uint8_t *roots_host,
                   uint64_t *alphas_host, uint64_t *last_host, size_t *last_len, const LeafSrc *round0_src) {
    SMI_TRY(smi_fri_check(ctx, cfg));
    if (cfg->domain_length != len) return smi_fail(ctx, SMI_ERR_CODEWORD_LEN, "initial codeword length does not match domain length");
    const uint32_t p = ctx->fs.F.p;
    if (cfg->omega >= p || cfg->offset >= p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "omega/offset must be < p");
    uint64_t R;
    smi_fri_num_rounds(cfg, &R);
    if (R == 0) return smi_fail(ctx, SMI_ERR_NO_ROUNDS, "num_rounds() == 0: the reference's verify rejects such a proof");
    const uint64_t t = cfg->num_colinearity_tests;
    const uint64_t last_n = len >> (R - 1);
    if (do_query) {  // asserts of src/fri.rs:183-192
        if (t > 2 * last_n) return smi_fail(ctx, SMI_ERR_SAMPLE_ENTROPY, "not enough entropy in indices wrt last codeword");
        if (t > last_n) return smi_fail(ctx, SMI_ERR_SAMPLE_TOO_MANY, "cannot sample more indices than available in last codeword");
    }

    ScaleScope pin__(ctx);   // the fused tail collects one x^-1 table per fold before its single launch
    smi_fri_run *run = new smi_fri_run();
    run->ctx = ctx;
    run->owns_first = false;
    run->arena = run_out == nullptr;   // nothing outlives the call: use the recycled arena
    if (run->arena && reset_arena) (void)arena_reset(ctx);
    run->d_misc = nullptr;
    run->d_proof = nullptr;
    int rc = SMI_OK;
    auto bail = [&](int code) {
        smi_fri_run_free(run);
        return code;
    };

    // proof layout (src/fri.rs:129,151,229-243; tags src/stream.rs:39-60)
    const size_t off_roots = 0, off_last = 33 * R, off_layers = off_last + 9 + 8 * last_n;
    std::vector<LayerInfo> layers(R ? R - 1 : 0);
    size_t off = off_layers;
    for (uint64_t i = 0; i + 1 < R; i++) {
        const uint64_t li = len >> i;
        const uint32_t d = ilog2(li);
        layers[i].len = li;
        layers[i].depth = d;
        layers[i].depth_next = d - 1;
        layers[i].off_triples = off;
        off += 33 * t;
        layers[i].off_paths = off;
        off += t * (2 * (9 + 32ull * d) + (9 + 32ull * (d - 1)));
    }
    const size_t proof_len = do_query ? off : off_layers;

    const MiscLayout ml = misc_layout(R, t);
    if (!(run->d_misc = run_alloc(run, ml.total))) return bail(smi_fail(ctx, SMI_ERR_OOM, "alloc misc"));
    // The challenges, the sampled indices and the caller's ride-along bytes live behind the proof in ONE device buffer, so
    // that everything the host needs comes back in a single device-to-host copy: four copies cost 17 + 5 + 4 + 4 us of blit
    // kernels and 86 us of runtime gaps between them at the end of every prove (profiles/r03_b_prove_timeline.txt).
    const size_t off_al = (proof_len + 7) & ~(size_t)7, off_top = off_al + 8 * R, off_ride = off_top + 8 * (t + 1);
    const void *ride_src = ctx->ride_src;
    const size_t ride_bytes = ride_src ? ctx->ride_bytes : 0;
    void *ride_dst = ctx->ride_dst;
    ctx->ride_src = nullptr;
    ctx->ride_bytes = 0;
    ctx->ride_dst = nullptr;
    const size_t back_len = off_ride + ride_bytes;
    if (!(run->d_proof = (uint8_t *)run_alloc(run, back_len))) return bail(smi_fail(ctx, SMI_ERR_OOM, "alloc proof"));
    uint8_t *misc = (uint8_t *)run->d_misc;
    FsState *d_fs = (FsState *)(misc + ml.fs);
    uint64_t *d_alphas = (uint64_t *)(run->d_proof + off_al);
    uint64_t *d_seed_ch = (uint64_t *)(misc + ml.seed_ch);
    uint64_t *d_top = (uint64_t *)(run->d_proof + off_top);
    uint64_t *d_reduced = (uint64_t *)(misc + ml.reduced);
    LayerInfo *d_layers = (LayerInfo *)(misc + ml.layers);

    fs_init_kernel<<<1, 64, 0, ctx->stream>>>(d_fs, (const uint8_t *)ride_src, run->d_proof + off_ride, ride_bytes);

    uint32_t omega = (uint32_t)cfg->omega, offset = (uint32_t)cfg->offset;
    const uint32_t *cur = d_codeword;
    uint64_t cur_len = len;
    // Codewords of at most this many elements finish in the fused tail launch (hash.hip, fri_tail_kernel).
    // Measured on MI355X (2^25-point prove): the FRI stage is 4.44..4.69 ms with the tail
    // off, from 512, from 1024 or from 2048 elements alike -- the differences are inside the run-to-run
    // spread of +-0.1 ms, because the tail's cost is the serial chain tree -> alpha -> fold, not launch
    // overhead.  Default: 512, the size the per-round path hands to a single workgroup anyway (above it
    // the per-round path spreads a tree's 64-leaf chunks over several CUs).  SMI_FRI_TAIL=<len> overrides
    // (0: never).
    const uint64_t tail_len = fri_tail_len();
    static const bool fold_in_tail = !(getenv("SMI_MERKLE_FUSE") && atoi(getenv("SMI_MERKLE_FUSE")) == 0);
    // The leaves of a round's tree can be computed by the launch that hashes them (LeafSrc, internal.h): the initial
    // codeword as the caller's weighted column sum (round0_src), every later one as the fold of the round before --
    // wherever the tree starts with the four-leaves-per-lane kernel (merkle_fuses_leaf_source).  The codeword buffer is
    // written by that launch; everything downstream reads it as before.
    LeafSrc pending;
    memset(&pending, 0, sizeof pending);
    bool have_pending = false;
    if (round0_src) {
        if (!merkle_fuses_leaf_source(len) || !(cur_len > tail_len)) return bail(smi_fail(ctx, SMI_ERR_BAD_ARG, "fri: the initial codeword is too short for a computed leaf source"));
        pending = *round0_src;
        have_pending = true;
    }
    for (uint64_t r = 0; r < R; r++) {
        if (cur_len <= tail_len && R - r <= SMI_FRI_TAIL_MAX_ROUNDS) {
            // every remaining round in one workgroup launch (hash.hip, fri_tail_kernel)
            // the tail holds one x^-1 table per fold until its launch: none of them may be evicted in between
            if ((rc = ctx_scale_reserve(ctx, (size_t)(R - r))) != SMI_OK) return bail(rc);
            FriTailArgs ta;
            memset(&ta, 0, sizeof ta);
            ta.n_rounds = (uint32_t)(R - r);
            ta.fs_words = d_fs->s;
            ta.F = ctx->fs.F;
            ta.inv2_m = (uint32_t)(((uint64_t)h_inv(ctx, 2) << 32) % p);
            if (have_pending) {   // the fold into this codeword runs at the head of the tail launch
                ta.pre_lo = pending.lo;
                ta.pre_hi = pending.hi;
                ta.pre_alpha = pending.alpha;
                ta.pre_S = pending.S;
                have_pending = false;
            }
            for (uint64_t k = r; k < R; k++) {
                FriTailRound &tr = ta.r[k - r];
                const bool last = k == R - 1;
                tr.cw = cur;
                tr.len = (uint32_t)cur_len;
                run->codewords.push_back(const_cast<uint32_t *>(cur));
                run->lens.push_back(cur_len);
                tr.nodes = (uint8_t *)run_alloc(run, (2 * cur_len - 1) * 32);
                if (!tr.nodes) return bail(smi_fail(ctx, SMI_ERR_OOM, "alloc tree"));
                run->trees.push_back(tr.nodes);
                tr.proof_slot = run->d_proof + off_roots + 33 * k;
                tr.alpha_out = last ? nullptr : d_alphas + k;
                if (last) break;
                if (cur_len < 2) return bail(smi_fail(ctx, SMI_ERR_BAD_ARG, "fold: codeword length must be a power of two >= 2"));
                if (offset == 0 || omega == 0) return bail(smi_fail(ctx, SMI_ERR_DIV_BY_ZERO, "no division by zero"));   // src/ff.rs:182
                if ((rc = ctx_scale_tables(ctx, h_inv(ctx, offset), h_inv(ctx, omega), ilog2(cur_len / 2), &tr.S)) != SMI_OK) return bail(rc);
                tr.next = (uint32_t *)run_alloc(run, (cur_len / 2) * 4);
                if (!tr.next) return bail(smi_fail(ctx, SMI_ERR_OOM, "alloc codeword"));
                cur = tr.next;
                cur_len /= 2;
                omega = h_mul(ctx, omega, omega);    // src/fri.rs:146-147
                offset = h_mul(ctx, offset, offset);
            }
            if ((rc = launch_fri_tail(ctx, ta)) != SMI_OK) return bail(rc);
            break;
        }
        // leaf hashes + tree (src/fri.rs:118-127); power-of-two lengths never need padding
        run->codewords.push_back(const_cast<uint32_t *>(cur));   // owned by the run from here on: nothing below can leak it
        run->lens.push_back(cur_len);
        uint8_t *nodes = (uint8_t *)run_alloc(run, (2 * cur_len - 1) * 32);
        if (!nodes) return bail(smi_fail(ctx, SMI_ERR_OOM, "alloc tree"));
        run->trees.push_back(nodes);
        const uint32_t *root = (const uint32_t *)(nodes + (2 * cur_len - 2) * 32);
        const bool last = r == R - 1;
        // push root, absorb, challenge (src/fri.rs:129-138): done by the workgroup that finishes the tree
        // when that is the chunk kernel (one launch fewer per round), by a kernel of its own otherwise
        bool fs_done = false;
        if (have_pending) {
            rc = launch_merkle_src_fs(ctx, pending, cur_len, nodes, d_fs->s, run->d_proof + off_roots + 33 * r, last ? nullptr : d_alphas + r, &fs_done);
            have_pending = false;
        } else {
            rc = launch_merkle_fs(ctx, cur, cur_len, nodes, d_fs->s, run->d_proof + off_roots + 33 * r, last ? nullptr : d_alphas + r, &fs_done);
        }
        if (rc != SMI_OK) return bail(rc);
        if (!fs_done)
            fs_round_kernel<<<1, 64, 0, ctx->stream>>>(d_fs, root, run->d_proof + off_roots + 33 * r, last ? nullptr : d_alphas + r);
        if (last) break;
        uint32_t *next = (uint32_t *)run_alloc(run, (cur_len / 2) * 4);
        if (!next) return bail(smi_fail(ctx, SMI_ERR_OOM, "alloc codeword"));
        const uint64_t next_len = cur_len / 2;
        const bool next_is_tail = next_len <= tail_len && R - (r + 1) <= SMI_FRI_TAIL_MAX_ROUNDS;
        // (the computed-leaf kernel reads and writes four elements at a time: 16-byte aligned buffers only -- the library's own
        // are, a caller's initial codeword need not be)
        const bool aligned16 = (((uintptr_t)cur | (uintptr_t)next) & 15u) == 0;
        if ((merkle_fuses_leaf_source(next_len) && !next_is_tail && aligned16) || (merkle_chunks_fold(next_len) && !next_is_tail) ||
            (next_is_tail && fold_in_tail)) {
            // no fold launch: the next round's leaf kernel (the four-leaves-per-lane kernel, the chunk kernel or the fused
            // tail, whichever that round starts with) folds (same checks and tables as launch_fold_shard)
            if (offset == 0 || omega == 0) {
                if (!run->arena) (void)hipFree(next);
                return bail(smi_fail(ctx, SMI_ERR_DIV_BY_ZERO, "no division by zero"));   // src/ff.rs:182
            }
            memset(&pending, 0, sizeof pending);
            pending.kind = LEAF_FOLD;
            pending.cw_out = next;
            pending.F = ctx->fs.F;
            pending.lo = cur;
            pending.hi = cur + next_len;
            pending.alpha = d_alphas + r;
            pending.inv2_m = (uint32_t)(((uint64_t)h_inv(ctx, 2) << 32) % p);
            if ((rc = ctx_scale_tables(ctx, h_inv(ctx, offset), h_inv(ctx, omega), ilog2(next_len), &pending.S)) != SMI_OK) {
                if (!run->arena) (void)hipFree(next);
                return bail(rc);
            }
            have_pending = true;
        } else if ((rc = launch_fold(ctx, cur, cur_len, d_alphas + r, offset, omega, next)) != SMI_OK) {
            if (!run->arena) (void)hipFree(next);
            return bail(rc);
        }
        cur = next;
        cur_len /= 2;
        omega = h_mul(ctx, omega, omega);    // src/fri.rs:146-147
        offset = h_mul(ctx, offset, offset);
    }
    // last codeword in the clear (src/fri.rs:151)
    emit_codeword_kernel<<<(uint32_t)((cur_len + 255) / 256), 256, 0, ctx->stream>>>(cur, cur_len, run->d_proof + off_last);

    if (do_query) {
        fs_challenge_kernel<<<1, 64, 0, ctx->stream>>>(d_fs, d_seed_ch);
        const uint64_t sample_size = R > 1 ? len / 2 : len;  // src/fri.rs:266-270
        sample_indices_kernel<<<1, 64, 0, ctx->stream>>>(d_seed_ch, sample_size, last_n, (uint32_t)t, d_top, d_reduced);
        if (R > 1 && t > 0) {
            for (uint64_t i = 0; i + 1 < R; i++) {
                layers[i].cw = run->codewords[i];
                layers[i].cw_next = run->codewords[i + 1];
                layers[i].nodes = run->trees[i];
                layers[i].nodes_next = run->trees[i + 1];
            }
            if (R - 1 <= SMI_QUERY_TAB_MAX) {
                LayerTable tab;
                memset(&tab, 0, sizeof tab);
                memcpy(tab.l, layers.data(), sizeof(LayerInfo) * (R - 1));
                query_tab_kernel<<<dim3((uint32_t)t, (uint32_t)(R - 1)), 64, 0, ctx->stream>>>(tab, d_top, (uint32_t)t, run->d_proof);
            } else {
                if (hipMemcpyAsync(d_layers, layers.data(), sizeof(LayerInfo) * (R - 1), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    return bail(smi_fail(ctx, SMI_ERR_HIP, "hipMemcpyAsync layers"));
                query_kernel<<<dim3((uint32_t)t, (uint32_t)(R - 1)), 64, 0, ctx->stream>>>(d_layers, d_top, (uint32_t)t, run->d_proof);
            }
        }
    }
    if (hipGetLastError() != hipSuccess) return bail(smi_fail(ctx, SMI_ERR_HIP, "fri kernel launch"));

    // one synchronising copy-back, through the context's pinned landing buffer: proof | challenges | indices | ride-along
    uint8_t *land = nullptr;
    if ((rc = ctx_pin_out(ctx, back_len, &land)) != SMI_OK) return bail(rc);
    if (hipMemcpyAsync(land, run->d_proof, back_len, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        return bail(smi_fail(ctx, SMI_ERR_HIP, "copy proof"));
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(smi_hip_fail(ctx, e, "fri sync"));
    std::vector<uint8_t> proof(land, land + proof_len);
    std::vector<uint64_t> alphas(R), top(t + 1);
    if (R > 1) memcpy(alphas.data(), land + off_al, 8 * (R - 1));
    if (do_query && t) memcpy(top.data(), land + off_top, 8 * t);
    if (ride_bytes && ride_dst) memcpy(ride_dst, land + off_ride, ride_bytes);

    if (roots_host)
        for (uint64_t r = 0; r < R; r++) memcpy(roots_host + 32 * r, proof.data() + 33 * r + 1, 32);
    if (alphas_host) memcpy(alphas_host, alphas.data(), 8 * (R - 1));
    if (last_host) memcpy(last_host, proof.data() + off_last + 9, 8 * last_n);
    if (last_len) *last_len = last_n;
    if (top_host && do_query) memcpy(top_host, top.data(), 8 * t);
    if (proof_host) proof_host->swap(proof);
    if (run_out) *run_out = run;
    else smi_fri_run_free(run);
    return SMI_OK;
}

// ------------------------------------------------------------------------- C ABI
// Accessors of a retained commit: the `codewords` Fri::commit returns (src/fri.rs:153-155) and
// MerkleTree::open on the per-round trees (src/fri.rs:297-298 rebuilds them; here they are kept).
int smi_fri_run_num_codewords(const smi_fri_run *run, size_t *n) {
    if (!run || !n) return SMI_ERR_BAD_ARG;
    *n = run->codewords.size();
    return SMI_OK;
}
int smi_fri_run_codeword(smi_fri_run *run, size_t round, uint64_t *out, size_t *len) {
    if (!run || !len) return SMI_ERR_BAD_ARG;
    if (round >= run->codewords.size()) return smi_fail(run->ctx, SMI_ERR_INDEX_OOB, nullptr);
    *len = run->lens[round];
    if (!out) return SMI_OK;
    return smi_dev_download_u64(run->ctx, run->codewords[round], run->lens[round], out);
}
int smi_fri_run_open(smi_fri_run *run, size_t round, size_t index, uint8_t *path, size_t *depth) {
    if (!run || !path || !depth) return SMI_ERR_BAD_ARG;
    if (round >= run->trees.size()) return smi_fail(run->ctx, SMI_ERR_INDEX_OOB, nullptr);
    smi_tree t{run->ctx, run->trees[round], (size_t)run->lens[round], false};
    return smi_merkle_open(run->ctx, &t, index, path, depth);
}

int smi_dev_fri_fold(smi_ctx *ctx, const uint32_t *d_in, size_t len, const uint64_t *d_alpha, uint64_t offset,
                     uint64_t omega, uint32_t *d_out) {
    if (!ctx || !d_in || !d_alpha || !d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return launch_fold(ctx, d_in, len, d_alpha, offset, omega, d_out);
}

int smi_dev_fri_prove(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint32_t *d_codeword, size_t len, uint8_t **proof,
                      size_t *proof_len, uint64_t *top_indices, smi_fri_run **run) {
    if (!ctx || !cfg || !d_codeword || !proof || !proof_len) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    std::vector<uint8_t> bytes;
    SMI_TRY(fri_run(ctx, cfg, d_codeword, len, true, true, run, &bytes, top_indices, nullptr, nullptr, nullptr, nullptr, nullptr));
    *proof = (uint8_t *)malloc(bytes.size() ? bytes.size() : 1);
    if (!*proof) return smi_fail(ctx, SMI_ERR_OOM, "malloc proof");
    memcpy(*proof, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    return SMI_OK;
}

static int upload_codeword(smi_ctx *ctx, const uint64_t *codeword, size_t len, uint32_t **d_cw) {
    if (hipMalloc((void **)d_cw, len * 4 ? len * 4 : 4) != hipSuccess) return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc codeword");
    return host_to_dev_u32(ctx, codeword, len, *d_cw, 0);
}

int smi_fri_prove(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint64_t *codeword, size_t len, uint8_t **proof,
                  size_t *proof_len, uint64_t *top_indices) {
    if (!ctx || !cfg || !codeword || !proof || !proof_len) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    SMI_TRY(smi_fri_check(ctx, cfg));
    if (cfg->domain_length != len) return smi_fail(ctx, SMI_ERR_CODEWORD_LEN, "initial codeword length does not match domain length");
    uint32_t *d_cw = nullptr;
    int rc = upload_codeword(ctx, codeword, len, &d_cw);
    if (rc == SMI_OK) rc = smi_dev_fri_prove(ctx, cfg, d_cw, len, proof, proof_len, top_indices, nullptr);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_cw);
    return rc;
}

int smi_fri_commit(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint64_t *codeword, size_t len, uint8_t *roots,
                   uint64_t *alphas, uint64_t *last_codeword, size_t *last_len, smi_fri_run **run) {
    if (!ctx || !cfg || !codeword) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    SMI_TRY(smi_fri_check(ctx, cfg));
    if (cfg->domain_length != len) return smi_fail(ctx, SMI_ERR_CODEWORD_LEN, "initial codeword length does not match domain length");
    uint32_t *d_cw = nullptr;
    int rc = upload_codeword(ctx, codeword, len, &d_cw);
    smi_fri_run *r = nullptr;
    if (rc == SMI_OK) rc = fri_run(ctx, cfg, d_cw, len, false, true, run ? &r : nullptr, nullptr, nullptr, roots, alphas, last_codeword, last_len, nullptr);
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == SMI_OK && run) {
        r->owns_first = true;  // the run keeps the uploaded codeword
        *run = r;
    } else {
        (void)hipFree(d_cw);
    }
    return rc;
}

int smi_fri_fold(smi_ctx *ctx, const uint64_t *codeword, size_t len, uint64_t alpha, uint64_t offset, uint64_t omega,
                 uint64_t *out) {
    if (!ctx || !codeword || !out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (len < 2 || !is_pow2(len)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "fold: codeword length must be a power of two >= 2");
    void *d_in, *d_out, *d_alpha;
    SMI_TRY(ctx_tmp(ctx, 1, len * 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, len * 2 + 8, &d_out));
    SMI_TRY(ctx_tmp(ctx, 3, 8, &d_alpha));
    SMI_TRY(host_to_dev_u32(ctx, codeword, len, (uint32_t *)d_in, 0));
    HIP_TRY(ctx, hipMemcpyAsync(d_alpha, &alpha, 8, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_fold(ctx, (const uint32_t *)d_in, len, (const uint64_t *)d_alpha, offset, omega, (uint32_t *)d_out));
    return dev_u32_to_host(ctx, (const uint32_t *)d_out, len / 2, out);
}
