// fourstep.hip -- local pieces of the multi-GPU four-step NTT (the reference has
// no distributed code at all).  The exchange step itself is an RCCL all-to-all issued by the
// host process (stark_rs_amd/fourstep.py); these kernels produce / consume its buffers.
#include "internal.h"

// element (cl, kr) of the column-major block times offset^c * w_N^(kr*c), c = c0 + cl, written
// to send[kr / rpg][cl][kr % rpg]  (rpg = R / n_ranks rows per destination rank).
// One workgroup walks a 4096-row stretch of one column: lane l takes rows kr0 + l + 256*j, so loads
// and stores are coalesced and a thread's twiddles g^kr (g = w_N^c) are a running product with ratio
// g^256 -- two table look-ups per thread instead of one scattered look-up per element (which left
// this kernel at 1.6 TB/s).  HBM-bound: 4 B in, 4 B out per element.
#define SMI_FS_THREADS 256
#define SMI_FS_PER 16
__global__ __launch_bounds__(SMI_FS_THREADS) void fourstep_twiddle_pack_kernel(const uint32_t *__restrict__ cols, uint32_t *__restrict__ send,
                                                                                uint32_t log_r, uint32_t log_n, uint32_t c0,
                                                                                uint32_t n_local, uint32_t rpg_log, Fp F, NttTables T,
                                                                                ScaleTables S, int use_scale) {
    const uint32_t R = 1u << log_r, span = SMI_FS_THREADS * SMI_FS_PER;
    const uint32_t spans_per_col = (R + span - 1) / span;
    const uint32_t cl = blockIdx.x / spans_per_col, kr0 = (blockIdx.x % spans_per_col) * span + threadIdx.x;
    if (cl >= n_local) return;
    const uint32_t c = c0 + cl, sh = T.K - log_n;
    // g^kr0 and g^256 (exponents mod N: the table index is e << sh with e < N)
    const uint32_t nmask = (uint32_t)((1ull << log_n) - 1ull);
    const uint32_t e0 = (uint32_t)(((uint64_t)kr0 * c) & nmask), es = (uint32_t)(((uint64_t)SMI_FS_THREADS * c) & nmask);
    uint32_t cur = e0 ? two_level(T.lo, T.hi, T.h, e0 << sh, F) : F.r1;
    const uint32_t ratio = es ? two_level(T.lo, T.hi, T.h, es << sh, F) : F.r1, rq = ratio * F.pinv;
    if (use_scale) cur = mont_mul(cur, two_level(S.lo, S.hi, S.h, c, F), F);   // offset^c rides in the base
    const uint32_t *col = cols + ((size_t)cl << log_r);
    uint32_t v[SMI_FS_PER];
#pragma unroll
    for (int j = 0; j < SMI_FS_PER; j++) {
        const uint32_t kr = kr0 + j * SMI_FS_THREADS;
        v[j] = kr < R ? col[kr] : 0u;
    }
#pragma unroll
    for (int j = 0; j < SMI_FS_PER; j++) {
        const uint32_t kr = kr0 + j * SMI_FS_THREADS;
        if (kr < R) {
            const uint32_t h = kr >> rpg_log, krl = kr & ((1u << rpg_log) - 1u);
            send[(((uint64_t)h * n_local + cl) << rpg_log) + krl] = mont_mul(v[j], cur, F);
        }
        cur = mont_mul_c(cur, ratio, rq, F);
    }
}

// out[c*rows + r] = in[r*cols + c]; 64x64 tiles through LDS (+1 padding: conflict-free)
__global__ __launch_bounds__(256) void transpose_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t rows, uint64_t cols) {
    __shared__ uint32_t tile[64][65];
    const uint64_t c0 = (uint64_t)blockIdx.x * 64, r0 = (uint64_t)blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (uint32_t j = ty; j < 64; j += 4) {
        const uint64_t r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) tile[j][tx] = in[r * cols + c];
    }
    __syncthreads();
    for (uint32_t j = ty; j < 64; j += 4) {
        const uint64_t c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][j];
    }
}

static uint32_t ilog2u(uint64_t n) {
    uint32_t l = 0;
    while ((n >> l) > 1) l++;
    return l;
}

int smi_dev_fourstep_twiddle_pack(smi_ctx *ctx, const uint32_t *d_cols, uint32_t *d_send, uint32_t log_r, uint32_t log_c,
                                  uint32_t c0, uint32_t n_local_cols, uint32_t n_ranks, int inverse, uint64_t offset) {
    if (!ctx || !d_cols || !d_send || !n_ranks) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    const uint32_t log_n = log_r + log_c, p = ctx->fs.F.p;
    if (log_n > ctx->fs.K) return smi_fail(ctx, p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "four-step size exceeds two-adicity");
    if ((n_ranks & (n_ranks - 1)) || ilog2u(n_ranks) > log_r) return smi_fail(ctx, SMI_ERR_BAD_ARG, "n_ranks must be a power of two <= R");
    if ((uint64_t)c0 + n_local_cols > (1ull << log_c)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "column block out of range");
    if (offset >= p || offset == 0) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "offset must be in [1,p)");
    ScaleTables S{nullptr, nullptr, 0};
    const int use_scale = offset != 1;
    if (use_scale) SMI_TRY(ctx_scale_tables(ctx, 1, (uint32_t)offset, log_c, &S));
    const uint64_t total = (uint64_t)n_local_cols << log_r;
    if (!total) return SMI_OK;
    const uint64_t span = (uint64_t)SMI_FS_THREADS * SMI_FS_PER, spans_per_col = ((1ull << log_r) + span - 1) / span;
    const uint64_t grid = spans_per_col * n_local_cols;
    if (grid > 0x7FFFFFFFull) return smi_fail(ctx, SMI_ERR_BAD_ARG, "four-step block too large");
    ProfScope ps(ctx, "fourstep_twiddle_pack_kernel", 8.0 * (double)total);
    fourstep_twiddle_pack_kernel<<<(uint32_t)grid, SMI_FS_THREADS, 0, ctx->stream>>>(d_cols, d_send, log_r, log_n, c0, n_local_cols,
                                                                          log_r - ilog2u(n_ranks), ctx->fs.F,
                                                                          ctx_tables(ctx, inverse), S, use_scale);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

int smi_dev_transpose(smi_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, size_t rows, size_t cols) {
    if (!ctx || !d_in || !d_out || d_in == d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (!rows || !cols) return SMI_OK;
    const dim3 grid((uint32_t)((cols + 63) / 64), (uint32_t)((rows + 63) / 64));
    if (grid.y > 65535) return smi_fail(ctx, SMI_ERR_BAD_ARG, "transpose: too many rows");
    ProfScope ps(ctx, "transpose_kernel", 8.0 * (double)rows * (double)cols);
    transpose_kernel<<<grid, 256, 0, ctx->stream>>>(d_in, d_out, rows, cols);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
