// stark.hip -- the build-defined composition "trace -> LDE -> commit -> combine -> Fri::prove".
//
// The reference has no prover that connects Trace to Fri::prove (no AIR, no
// quotient, no LDE function).  This file composes the reference's primitives in the way
// SURVEY 8(d) cfg5 specifies, entirely on the device:
//   1. per-column iNTT on the trace domain + coset NTT on the blowup domain (ntt.hip);
//   2. one Merkle tree per column, one element per leaf -- the reference's only leaf rule
//      (src/fri.rs:118-121); or, with cfg.row_leaves, one tree whose leaf i hashes row i of the
//      extended trace (Hash::from_field_elements(&row), a build-defined variant: four columns are one
//      32-byte chunk, so the commit costs a quarter);
//   3. a fresh FiatShamir absorbs the column roots in order and draws one weight per column
//      (src/fiat_shamir.rs:15-25); codeword = sum_c weight_c * column_c;
//   4. Fri::prove semantics on that codeword (src/fri.rs:250-311), fresh FiatShamir as in the
//      reference's own tests (src/fri.rs:543-547).
#include <vector>

#include "hash_core.h"
#include "hash_hex.h"
#include "internal.h"
#include "mgpu_core.h"

int launch_merkle_batch(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t n_trees, size_t elem_stride,
                        size_t node_stride_bytes, uint32_t row_cols = 0, size_t row_stride = 0);
int launch_merkle_rows(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t col_stride, size_t n, uint8_t *d_nodes);
int fri_run(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint32_t *d_codeword, size_t len, bool do_query, bool reset_arena,
            smi_fri_run **run_out, std::vector<uint8_t> *proof_host, uint64_t *top_host, uint8_t *roots_host,
            uint64_t *alphas_host, uint64_t *last_host, size_t *last_len, const LeafSrc *round0_src);

// weights[c] = FiatShamir::challenge after absorbing roots[0..c] (unreduced u64)
// weights_m (optional): the same weights reduced mod p in Montgomery form, what the fused combination multiplies by
// the roots are named by a pointer table, or (root_ptrs == nullptr) sit root_stride bytes apart from root0 on
__global__ void fs_weights_kernel(const uint8_t *const *root_ptrs, uint32_t n, uint64_t *weights, uint32_t *roots_out,
                                  uint32_t *weights_m = nullptr, Fp F = Fp{0, 0, 0, 0}, const uint8_t *root0 = nullptr, size_t root_stride = 0) {
    if (threadIdx.x >= 16 || blockIdx.x) return;   // one state word per lane of the first row (hash_hex.h)
    const hashx::Lane row = hashx::make_lane(threadIdx.x);
    const uint32_t w = threadIdx.x, j = w >> 2;
    uint32_t x = row.init;
    for (uint32_t c = 0; c < n; c++) {
        const uint32_t *root = (const uint32_t *)(root_ptrs ? root_ptrs[c] : root0 + (size_t)c * root_stride);
        if (w < 8) roots_out[8 * c + w] = root[w];   // gathered: one copy to the host
        x = hashx::fs_absorb(x, hashx::message(root[j], root[4 + j], row), row);
        const uint64_t wt = hashx::low_bytes_u64(hashx::fs_challenge(x, row));
        if (w == 0) {
            weights[c] = wt;
            if (weights_m) weights_m[c] = to_mont_u64(wt, F);
        }
    }
}

// column openings (mgpu_core.h): one workgroup per (test, column)
__global__ __launch_bounds__(64) void column_open_kernel(const MgSide *cols, uint32_t W, const uint64_t *top, uint32_t t, int rank, uint8_t *out) {
    mg_column_open_write(cols, W, blockIdx.y, top[blockIdx.x], blockIdx.x, t, rank, out, threadIdx.x, 64);
}
int launch_column_open(smi_ctx *ctx, const MgSide *d_cols, uint32_t W, const uint64_t *d_top, uint32_t t, int rank, uint8_t *d_out) {
    if (!W || !t) return SMI_OK;
    column_open_kernel<<<dim3(t, W), 64, 0, ctx->stream>>>(d_cols, W, d_top, t, rank, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

int launch_fs_weights(smi_ctx *ctx, const uint8_t *const *d_root_ptrs, uint32_t n, uint64_t *weights, uint8_t *roots_out) {
    fs_weights_kernel<<<1, 64, 0, ctx->stream>>>(d_root_ptrs, n, weights, (uint32_t *)roots_out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

// row-leaf variant: a single root enters the transcript; weight c = FiatShamir::challenge of the
// transcript root || c as LE u64 (absorb(root); absorb(c.to_le_bytes()); challenge() on a clone)
__global__ void fs_row_weights_kernel(const uint8_t *root, uint32_t n, uint64_t *weights, uint8_t *roots_out, uint32_t *weights_m = nullptr,
                                      Fp F = Fp{0, 0, 0, 0}) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    uint8_t msg[40];
    for (int i = 0; i < 32; i++) msg[i] = root[i];
    if (c == 0)
        for (int i = 0; i < 32; i++) roots_out[i] = msg[i];
    for (int i = 0; i < 8; i++) msg[32 + i] = i < 4 ? (uint8_t)(c >> (8 * i)) : 0;
    uint32_t d[8];
    hashc::hash_bytes(msg, 40, d);
    weights[c] = (uint64_t)d[0] | ((uint64_t)d[1] << 32);
    if (weights_m) weights_m[c] = to_mont_u64(weights[c], F);
}

// out[i] = sum_c (weights[c] mod p) * cols[c*stride + i]      HBM-bound: 4*(n_cols+1) B per element
__global__ __launch_bounds__(256) void combine_columns_kernel(const uint32_t *__restrict__ cols, uint32_t n_cols, size_t len,
                                                              size_t stride, const uint64_t *__restrict__ weights, Fp F,
                                                              uint32_t *__restrict__ out) {
    __shared__ uint32_t w_m[64];
    if (threadIdx.x < n_cols) w_m[threadIdx.x] = to_mont((uint32_t)(weights[threadIdx.x] % F.p), F);
    __syncthreads();
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += step) {
        uint32_t acc = 0;
        for (uint32_t c = 0; c < n_cols; c++) acc = fp_add(acc, mont_mul(cols[c * stride + i], w_m[c], F), F.p);
        out[i] = acc;
    }
}

int smi_dev_combine_columns(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t len, size_t stride,
                            const uint64_t *d_weights, uint32_t *d_out) {
    if (!ctx || !d_cols || !d_weights || !d_out || !n_cols) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (n_cols > 64) return smi_fail(ctx, SMI_ERR_BAD_ARG, "combine: at most 64 columns per call");
    if (!len) return SMI_OK;
    size_t grid = (len + 255) / 256;
    if (grid > 4096) grid = 4096;
    ProfScope ps(ctx, "combine_columns_kernel", 4.0 * (n_cols + 1.0) * (double)len);
    combine_columns_kernel<<<(uint32_t)grid, 256, 0, ctx->stream>>>(d_cols, n_cols, len, stride, d_weights, ctx->fs.F, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

int smi_dev_stark_prove(smi_ctx *ctx, const smi_stark_cfg *cfg, const uint32_t *d_trace_cols, uint8_t *column_roots,
                        uint8_t **proof, size_t *proof_len, uint64_t *top_indices, double *stage_ms) {
    if (!ctx || !cfg || !d_trace_cols || !proof || !proof_len) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    const uint32_t W = cfg->n_cols, log_N = cfg->log_n + cfg->log_blowup;
    if (!W || W > 64) return smi_fail(ctx, SMI_ERR_BAD_ARG, "stark_prove: 1..64 columns");
    if (cfg->log_blowup < 2) return smi_fail(ctx, SMI_ERR_EXPANSION_TOO_SMALL, nullptr);  // Fri::new, src/fri.rs:45
    if (log_N > ctx->fs.K)
        return smi_fail(ctx, ctx->fs.F.p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "LDE domain too large");
    const size_t N = (size_t)1 << log_N;
    const uint32_t T = cfg->row_leaves ? 1u : W;   // trees (and roots entering the transcript)
    SMI_TRY(arena_reset(ctx));
    struct Events {   // destroyed on every return path
        hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Events() {
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } evs;
    hipEvent_t *ev = evs.ev;
    const bool timed = stage_ms != nullptr;
    if (timed)
        for (int i = 0; i < 5; i++) HIP_TRY(ctx, hipEventCreate(&ev[i]));
    auto mark = [&](int i) { if (timed) (void)hipEventRecord(ev[i], ctx->stream); };

    uint32_t *d_lde = (uint32_t *)arena_alloc(ctx, (size_t)W * N * 4);
    uint32_t *d_cw = (uint32_t *)arena_alloc(ctx, N * 4);
    uint64_t *d_weights = (uint64_t *)arena_alloc(ctx, 8 * W);
    uint32_t *d_weights_m = (uint32_t *)arena_alloc(ctx, 4 * W);
    if (!d_lde || !d_cw || !d_weights || !d_weights_m) return smi_fail(ctx, SMI_ERR_OOM, "stark_prove: device memory");
    std::vector<uint8_t *> trees(W);
    std::vector<const uint8_t *> rootp(W);
    // the W column trees sit back to back (stride 2N digests) so one set of launches builds them all
    const size_t tree_stride = 2 * N * 32;
    uint8_t *tree_base = (uint8_t *)arena_alloc(ctx, tree_stride * T);
    if (!tree_base) return smi_fail(ctx, SMI_ERR_OOM, "stark_prove: tree memory");
    for (uint32_t c = 0; c < T; c++) {
        trees[c] = tree_base + c * tree_stride;
        rootp[c] = trees[c] + (2 * N - 2) * 32;
    }
    mark(0);
    SMI_TRY(smi_dev_lde(ctx, d_trace_cols, W, cfg->log_n, cfg->log_blowup, cfg->trace_offset, cfg->lde_offset, d_lde));
    mark(1);
    if (cfg->row_leaves) SMI_TRY(launch_merkle_rows(ctx, d_lde, W, N, N, tree_base));
    else SMI_TRY(launch_merkle_batch(ctx, d_lde, N, tree_base, W, N, tree_stride));
    mark(2);
    uint8_t *d_roots = (uint8_t *)arena_alloc(ctx, 32 * (size_t)T);
    if (!d_roots) return smi_fail(ctx, SMI_ERR_OOM, "stark_prove: device memory");
    if (cfg->row_leaves) {
        fs_row_weights_kernel<<<1, 64, 0, ctx->stream>>>(rootp[0], W, d_weights, d_roots, d_weights_m, ctx->fs.F);
    } else {
        // (no pointer table: a host-to-device copy here would sit between the commit and the first FRI launch)
        fs_weights_kernel<<<1, 64, 0, ctx->stream>>>(nullptr, W, d_weights, (uint32_t *)d_roots, d_weights_m, ctx->fs.F, rootp[0], tree_stride);
    }
    HIP_TRY(ctx, hipGetLastError());
    // Sum_c weight_c * col_c: by a kernel of its own, or -- when the first FRI tree starts with the four-leaves-per-lane
    // kernel -- by that kernel as it hashes (LeafSrc, internal.h): one launch and one pass over the codeword less
    LeafSrc csrc;
    memset(&csrc, 0, sizeof csrc);
    const bool fuse_combine = merkle_fuses_leaf_source(N) && W <= SMI_LEAF_COMBINE_MAX && N > fri_tail_len() &&
                              (((uintptr_t)d_lde | (uintptr_t)d_cw) & 15u) == 0;   // four elements per access, N is a multiple of 4
    if (fuse_combine) {
        csrc.kind = LEAF_COMBINE;
        csrc.cw_out = d_cw;
        csrc.F = ctx->fs.F;
        csrc.cols = d_lde;
        csrc.stride = N;
        csrc.n_cols = W;
        csrc.weights_m = d_weights_m;
    } else {
        SMI_TRY(smi_dev_combine_columns(ctx, d_lde, W, N, N, d_weights, d_cw));
    }
    mark(3);
    smi_fri_cfg fc;
    fc.omega = h_root(ctx, log_N);
    fc.offset = cfg->lde_offset;
    fc.domain_length = N;
    fc.expansion_factor = 1ull << cfg->log_blowup;
    fc.num_colinearity_tests = cfg->num_colinearity_tests;
    std::vector<uint8_t> bytes;
    // the column openings need the top-level indices whether or not the caller wants them back
    std::vector<uint64_t> top_tmp(top_indices ? 0 : (size_t)cfg->num_colinearity_tests);
    if (!top_indices && !top_tmp.empty()) top_indices = top_tmp.data();
    if (column_roots) {   // the roots come back with fri_run's one copy-back instead of a copy and a sync of their own
        ctx->ride_src = d_roots;
        ctx->ride_bytes = 32 * (size_t)T;
        ctx->ride_dst = column_roots;
    }
    const int fri_rc = fri_run(ctx, &fc, d_cw, N, true, false, nullptr, &bytes, top_indices, nullptr, nullptr, nullptr, nullptr,
                               fuse_combine ? &csrc : nullptr);
    ctx->ride_src = nullptr;   // not consumed if fri_run failed early
    ctx->ride_bytes = 0;
    ctx->ride_dst = nullptr;
    SMI_TRY(fri_rc);
    mark(4);
    if (cfg->open_columns && !cfg->row_leaves && cfg->num_colinearity_tests) {
        // the top-level indices are on the host now (fri_run synchronised): one more small launch
        const uint32_t t = (uint32_t)cfg->num_colinearity_tests;
        std::vector<MgSide> sides(W);
        for (uint32_t c = 0; c < W; c++) {
            MgSide &sd = sides[c];
            sd.cw = d_lde + (size_t)c * N; sd.nodes = trees[c]; sd.top = nullptr; sd.len = N; sd.blk = N; sd.depth_local = log_N; sd.depth_top = 0;
        }
        const size_t ob = (size_t)mg_column_open_bytes(W, t, log_N);
        MgSide *d_sides = (MgSide *)arena_alloc(ctx, sizeof(MgSide) * W);
        uint64_t *d_top2 = (uint64_t *)arena_alloc(ctx, 8 * (size_t)t);
        uint8_t *d_open = (uint8_t *)arena_alloc(ctx, ob);
        if (!d_sides || !d_top2 || !d_open) return smi_fail(ctx, SMI_ERR_OOM, "stark_prove: column openings");
        HIP_TRY(ctx, hipMemcpyAsync(d_sides, sides.data(), sizeof(MgSide) * W, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_top2, top_indices, 8 * (size_t)t, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(d_open, 0, ob, ctx->stream));
        SMI_TRY(launch_column_open(ctx, d_sides, W, d_top2, t, 0, d_open));
        const size_t at = bytes.size();
        bytes.resize(at + ob);
        HIP_TRY(ctx, hipMemcpyAsync(bytes.data() + at, d_open, ob, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (timed) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 4; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            stage_ms[i] = ms;
        }
    }
    *proof = (uint8_t *)malloc(bytes.size() ? bytes.size() : 1);
    if (!*proof) return smi_fail(ctx, SMI_ERR_OOM, "malloc proof");
    memcpy(*proof, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    return SMI_OK;
}
