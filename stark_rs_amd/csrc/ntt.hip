// ntt.hip -- gfx950 NTT kernels and their host driver.
//
// Replaces Polynomial::interpolate_domain (reference src/univariate/interpolate.rs:6-44),
// Polynomial::eval_domain (src/univariate/eval.rs:16-21) and Polynomial::scale
// (src/univariate/mod.rs:99-113) on geometric domains.  Kernel structure, tiling and the
// roofline that bounds it are described in ntt_core.h and DESIGN.md.
#include "internal.h"
#include "ntt_driver.h"
#include "lde_core.h"

// ------------------------------------------------------------------------- kernels
// One workgroup per (tile, column).  Tried and dropped (r01, measured on MI355X): persistent
// workgroups that prefetch the next tile's loads into registers -- the extra live registers cost
// more occupancy than the prefetch wins (2^25 x 4 middle pass 258 -> 290 us), and the hardware
// dispatcher already staggers the load / compute / store phases of co-resident workgroups.
template <int LOGR, int LOGW, int KIND, int CAP>
__global__ __launch_bounds__(1 << (LOGR + LOGW - 4)) void ntt_pass_kernel(const PassArgs a) {
    typedef NttPass<LOGR, LOGW, KIND, CAP> NP;
    __shared__ uint32_t tile[NP::R * NP::WP];
    __shared__ Tw2 tw[NP::R];
    const uint32_t tid = threadIdx.x, batch = blockIdx.y;
    const typename NP::TileId t = NP::tile_id(a, blockIdx.x);
    NP::load_tw(a, tw, tid);
    uint32_t v[NP::V];
    if constexpr (KIND == PASS_FIRST) {
        // zero-padded first passes (an LDE's blowup) skip the loads and the degenerate butterfly
        // stages of the padding; zlog is wave-uniform
#define ZCASE(Z)                                         \
    case Z:                                              \
        NP::template load_regs<Z>(a, t, batch, v, tid);  \
        __syncthreads(); /* twiddle table staged */      \
        NP::template step0_regs<Z>(a, v, tile, tw, tid); \
        break;
        switch (a.zlog) { ZCASE(2) ZCASE(3) ZCASE(4) default: ZCASE(0) }
#undef ZCASE
    } else if constexpr (KIND == PASS_MID) {
        NP::template load_regs<0>(a, t, batch, v, tid);
        __syncthreads();
        NP::template step0_regs<0>(a, v, tile, tw, tid);
    } else if (a.flags & NTT_LAST_DIRECT) {
        NP::load_rows_direct(a, t, batch, v, tid);
        __syncthreads(); /* twiddle table staged */
        NP::step0_rows(a, v, tile, tw, tid);
    } else {
        NP::load_rows(a, t, batch, v, tid);
        NP::rows_to_lds(v, tile, tid);
        __syncthreads();
        NP::step0_lds(a, tile, tw, tid);
    }
    __syncthreads();
    if (NP::St::n == 3) {
        NP::step_mid(a, tile, tw, tid);
        __syncthreads();
    }
    NP::last_step_store(a, t, batch, tile, tw, tid);
}

// The same tile program with up to COLS_PER_WG columns of the tile taken by one workgroup (grid.y = the column groups):
// the output multipliers of a thread are derived once and applied to all of them (NttPass::out_mul).  The launcher
// picks it by NttPass::share_cols.
// Waves per SIMD the register allocation must leave room for, by log2 of the tile (tools/exp_share_cols.sh, 2^25 x 4
// middle pass): unbounded 81 VGPRs 229 us, 6 waves (73 VGPRs) 222 us, 8 waves (64 VGPRs, 8 spilled) 254 us; one
// workgroup per (tile, column) 237 us.
#ifndef SMI_COLS_WAVES
#define SMI_COLS_WAVES(TILE_LOG, LOGR) ((TILE_LOG) >= 14 || (LOGR) >= 11 ? 4 : ((LOGR) == 10 ? 5 : 6))   // what LDS and the allocator leave room for with 1024- and 2048-point lines
#endif
#ifndef SMI_COLS_TWIN_WAVES   // tuning builds: the bound and the companions of the variant that also holds input multipliers
#define SMI_COLS_TWIN_WAVES 4
#endif
#ifndef SMI_COLS_TWIN_MQ
#define SMI_COLS_TWIN_MQ 1
#endif
#ifndef SMI_COLS_PREFETCH   // the middle pass's column loop issues the next column's loads under the current column's last step and stores
#define SMI_COLS_PREFETCH 1  // (r03, profiles/r03_r_prefetch_ab.log, r03_s_prefetch_ab.log: -1.2 ... -1.8 % per step)
#endif
template <int LOGR, int LOGW, int KIND, int CAP, bool TWIN = false>   // TWIN: the previous pass left its twiddles to this one (NTT_TW_IN)
__global__ __launch_bounds__(1 << (LOGR + LOGW - 4), TWIN ? SMI_COLS_TWIN_WAVES : SMI_COLS_WAVES(LOGR + LOGW, LOGR)) void ntt_pass_cols_kernel(const PassArgs a) {
    typedef NttPass<LOGR, LOGW, KIND, CAP> NP;
    __shared__ uint32_t tile[NP::R * NP::WP];
    __shared__ Tw2 tw[NP::R];
    const uint32_t tid0 = threadIdx.x;
    const typename NP::TileId t = NP::tile_id(a, blockIdx.x);
    NP::load_tw(a, tw, tid0);
    constexpr bool MQ = !TWIN || SMI_COLS_TWIN_MQ;
    uint32_t mw[NP::V], mq[MQ ? NP::V : 1];
    NP::template out_mul<MQ>(a, t, tid0, mw, mq);
    uint32_t iw[TWIN ? NP::V : 1];
    if constexpr (TWIN) NP::in_mul(a, t, tid0, iw);
    const uint32_t col0 = blockIdx.y * NP::COLS_PER_WG, col1 = min(a.batch, col0 + (uint32_t)NP::COLS_PER_WG);
    // Software pipeline over the columns of the tile (middle passes with 32 lines per tile: the shapes of the 2^21+ plans,
    // where it fits the register bound without spilling; the other middle shapes would spill 2..7 registers and keep the
    // plain loop): once step 0 has moved a column's 16 values from the load registers into LDS those registers are free,
    // so the NEXT column's loads are issued there and fly under this column's last step and stores (the compiler places the
    // wait before their first use, at the top of the next iteration).  Same instructions, same results; only the order of
    // issue changes.  r03, three A/B rounds on one box (profiles/r03_s_prefetch_ab.log): 0.7014-0.7062 -> 0.6955-0.6971 ms
    // per step; giving the forward transform's last pass (no output scale, one column per workgroup) the same column loop
    // changes nothing further (0.6964-0.6974) and is not built.
    if constexpr (SMI_COLS_PREFETCH && KIND == PASS_MID && LOGW == 5) {
        uint32_t v[NP::V];
        auto load = [&](uint32_t col) {
            uint32_t tid = tid0;
            asm volatile("" : "+v"(tid));   // see below: keeps the tile program's addresses from being hoisted
            NP::template load_regs<0, !TWIN>(a, t, col, v, tid);
        };
        load(col0);
        for (uint32_t batch = col0; batch < col1; batch++) {
            uint32_t tid = tid0;
            asm volatile("" : "+v"(tid));
            if constexpr (TWIN) {
#pragma unroll
                for (int i = 0; i < NP::V; i++) v[i] = mont_mul(v[i], iw[i], a.F);
            }
            __syncthreads();                // the previous column's LDS reads are over (first column: the twiddle table is staged)
            NP::template step0_regs<0>(a, v, tile, tw, tid);
            if (batch + 1 < col1) load(batch + 1);
            __syncthreads();
            if (NP::St::n == 3) {
                NP::step_mid(a, tile, tw, tid);
                __syncthreads();
            }
            NP::template last_step_store_mul<MQ>(a, t, batch, tile, tw, tid, mw, mq);
        }
        return;
    }
    for (uint32_t batch = col0; batch < col1; batch++) {
        // Opaque per-iteration copy of the thread index: without it every address of the tile program (loop-invariant
        // across the columns) is hoisted and held in registers -- 150 VGPRs, or spills under a bound
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));
        uint32_t v[NP::V];
        // the barrier after the loads also separates the previous column's LDS reads from this column's writes
        static_assert(KIND != PASS_FIRST, "first passes keep one workgroup per (tile, column), NttPass::share_cols");
        if constexpr (KIND == PASS_MID) {
            NP::template load_regs<0, !TWIN>(a, t, batch, v, tid);
            if constexpr (TWIN) {
#pragma unroll
                for (int i = 0; i < NP::V; i++) v[i] = mont_mul(v[i], iw[i], a.F);
            }
            __syncthreads();
            NP::template step0_regs<0>(a, v, tile, tw, tid);
        } else if (a.flags & NTT_LAST_DIRECT) {
            NP::load_rows_direct(a, t, batch, v, tid);
            __syncthreads();
            NP::step0_rows(a, v, tile, tw, tid);
        } else {
            NP::load_rows(a, t, batch, v, tid);
            if (batch != col0) __syncthreads();
            NP::rows_to_lds(v, tile, tid);
            __syncthreads();
            NP::step0_lds(a, tile, tw, tid);
        }
        __syncthreads();
        if (NP::St::n == 3) {
            NP::step_mid(a, tile, tw, tid);
            __syncthreads();
        }
        NP::template last_step_store_mul<MQ>(a, t, batch, tile, tw, tid, mw, mq);
    }
}

// Measurement aid (smi_ctx_copy_probe): the same tile, the same global loads and the same store
// addresses as ntt_pass_kernel, no arithmetic and no LDS -- what HBM delivers for this pass's
// access pattern.  bench.py reports a pass's time against it beside the 8 TB/s peak.
template <int LOGR, int LOGW, int KIND>
__global__ __launch_bounds__(1 << (LOGR + LOGW - 4)) void ntt_copy_probe_kernel(const PassArgs a) {
    typedef NttPass<LOGR, LOGW, KIND, 4> NP;
    enum { NB = (NP::TILE / NP::RL) / NP::NT, KSTEP_LOG = LOGR - NP::SL };
    const uint32_t tid = threadIdx.x, batch = blockIdx.y;
    const typename NP::TileId t = NP::tile_id(a, blockIdx.x);
    uint32_t v[NP::V];
    // Layout experiment (SMI_PROBE_LINEAR, development only): what the pass would cost if its tile were one contiguous run
    // in the inter-pass buffer -- bit 0: loads linear, bit 1: stores linear (the other side keeps the pass's own pattern)
    const uint32_t lin = a.flags >> 30;
    const uint64_t lin_in = (uint64_t)batch * a.in_stride + (uint64_t)blockIdx.x * NP::TILE;
    const uint64_t lin_out = (uint64_t)batch * a.out_stride + (uint64_t)blockIdx.x * NP::TILE;
    if ((lin & 1u) && KIND != PASS_FIRST) {
#pragma unroll
        for (int i = 0; i < NP::V; i++) v[i] = ld32(a.in + lin_in, (uint32_t)i * NP::NT + tid);
    } else if constexpr (KIND == PASS_LAST) {
        if (a.flags & NTT_LAST_DIRECT) NP::load_rows_direct(a, t, batch, v, tid);
        else NP::load_rows(a, t, batch, v, tid);
    } else if constexpr (KIND == PASS_MID) {
        NP::template load_regs<0, false>(a, t, batch, v, tid);   // the loads only, also when the pass applies deferred twiddles
    } else {
        switch (a.zlog) {
        case 2: NP::template load_regs<2>(a, t, batch, v, tid); break;
        case 3: NP::template load_regs<3>(a, t, batch, v, tid); break;
        case 4: NP::template load_regs<4>(a, t, batch, v, tid); break;
        default: NP::template load_regs<0>(a, t, batch, v, tid); break;
        }
    }
    if ((lin & 2u) && KIND != PASS_LAST) {
#pragma unroll
        for (int i = 0; i < NP::V; i++) st32(a.out + lin_out, (uint32_t)i * NP::NT + tid, v[i] + 1u);
        return;
    }
    uint32_t *out = a.out + (uint64_t)batch * a.out_stride + t.out_base;
    const uint32_t sh = KIND == PASS_LAST ? a.Sp : a.L - a.Sp - LOGR;   // stride of the frequency index in the output
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        const uint32_t u = tid + bi * NP::NT;
        const uint32_t o0 = (NP::blk_to_k(u >> LOGW) << sh) + (u & (NP::W - 1));
#pragma unroll
        for (int kk = 0; kk < NP::RL; kk++) st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + sh)), v[bi * NP::RL + kk] + 1u);
    }
}

// Two-pass low-degree extension (lde_core.h): coefficient regrouping, pass A per (input tile, coset),
// pass B per 16 lines.
__global__ __launch_bounds__(256) void lde_coef_tile_kernel(const LdeArgs a) {
    __shared__ uint32_t tile[LdeCoefTile::T * (LdeCoefTile::T + 1) * 4];
    LdeCoefTile::load(a, blockIdx.x, blockIdx.y, blockIdx.z, tile, threadIdx.x);
    __syncthreads();
    LdeCoefTile::store(a, blockIdx.x, blockIdx.y, blockIdx.z, tile, threadIdx.x);
}
template <int LOGR, int CAP>
__global__ __launch_bounds__(1 << (LOGR - 2)) void lde_a_kernel(const LdeArgs a) {
    typedef LdeA<LOGR, CAP> A;
    __shared__ uint32_t tile[A::R * 4];
    __shared__ Tw2 tw[A::TWS];
    const uint32_t tid = threadIdx.x, batch = blockIdx.y;
    const typename A::TileId t = A::tile_id(a, blockIdx.x);
    A::load_tw(a, tw, tid);
    uint32_t v[A::V];
    A::load_regs(a, t, batch, v, tid);
    __syncthreads();
    A::step0(a, v, tile, tw, tid);
    __syncthreads();
    A::step_mid(a, tile, tw, tid);
    __syncthreads();
    A::last_step_store(a, t, batch, tile, tw, tid);
}
template <int CAP>
__global__ __launch_bounds__(1024) void lde_b_kernel(const LdeArgs a) {
    typedef LdeB<CAP> B;
    __shared__ uint32_t tile[B::R * B::WP];
    __shared__ Tw2 tw[B::R];
    const uint32_t tid = threadIdx.x, batch = blockIdx.y;
    const typename B::TileId t = B::tile_id(a, blockIdx.x);
    B::load_tw(a, tw, tid);
    uint32_t v[B::V];
    B::load(a, t, batch, v, tid);
    B::to_lds(a, v, tile, tid);
    __syncthreads();
    B::step0(a, t, tile, tw, tid);
    __syncthreads();
    B::step_mid(a, tile, tw, tid);
    __syncthreads();
    B::last_step_store(a, t, batch, tile, tw, tid);
}

template <int LOGR> __global__ __launch_bounds__(1 << (LOGR - 2)) void lde_a_probe_kernel(const LdeArgs a) {
    LdeAProbe<LOGR>::run(a, blockIdx.x, blockIdx.y, threadIdx.x);
}
__global__ __launch_bounds__(1024) void lde_b_probe_kernel(const LdeArgs a) { LdeBProbe::run(a, blockIdx.x, blockIdx.y, threadIdx.x); }

__global__ __launch_bounds__(SMI_NTT_THREADS) void ntt_small_kernel(const SmallArgs a) {
    __shared__ uint32_t buf[SMI_TILE + SMI_TILE / 64];
    __shared__ Tw2 twm[SMI_TILE / 2];
    const uint32_t tid = threadIdx.x, batch = blockIdx.x;
    NttSmall::load_tw(a, twm, tid);
    NttSmall::load(a, batch, buf, tid);
    __syncthreads();
    uint32_t s = 0;
    for (; s + 2 <= a.L; s += 2) {
        NttSmall::stage4(a, s, buf, twm, tid);
        __syncthreads();
    }
    if (s < a.L) {
        NttSmall::stage(a, s, buf, twm, tid);
        __syncthreads();
    }
    NttSmall::store(a, batch, buf, tid);
}

__global__ void geom_table_kernel(uint32_t *out, GeomSpec s, Fp F) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.count) return;
    const uint32_t v = geom_entry(s.c_m, s.q_m, i, s.stride, F);
    if (s.pair) {
        const Tw2 c = tw2_from_mont(v, F);
        out[2 * i] = c.w;
        out[2 * i + 1] = c.q;
    } else {
        out[i] = v;
    }
}

// u64 (reference wire width, src/stream.rs:45) <-> u32 device residues
__global__ void narrow_kernel(const uint64_t *in, uint32_t *out, size_t n, uint32_t p, int reduce, int *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (; i < n; i += step) {
        const uint64_t v = in[i];
        if (v >= p) bad = 1;
        out[i] = (uint32_t)(v % p);
    }
    if (bad && !reduce) *flag = 1;
}
__global__ void widen_kernel(const uint32_t *in, uint64_t *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = in[i];
}

// ------------------------------------------------------------------------- launches
static uint32_t grid_for(size_t n, uint32_t block) {
    size_t g = (n + block - 1) / block;
    return (uint32_t)(g > 2048 ? 2048 : (g ? g : 1));
}

int launch_geom_table(smi_ctx *ctx, const GeomSpec &s, uint32_t *d_out) {
    geom_table_kernel<<<(s.count + 255) / 256, 256, 0, ctx->stream>>>(d_out, s, ctx->fs.F);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_narrow(smi_ctx *ctx, const uint64_t *d_in, uint32_t *d_out, size_t n, int reduce) {
    if (!n) return SMI_OK;
    narrow_kernel<<<grid_for(n, 256), 256, 0, ctx->stream>>>(d_in, d_out, n, ctx->fs.F.p, reduce, ctx->d_flag);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_widen(smi_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, size_t n) {
    if (!n) return SMI_OK;
    widen_kernel<<<grid_for(n, 256), 256, 0, ctx->stream>>>(d_in, d_out, n);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

namespace {
struct HipLauncher {
    smi_ctx *ctx;
    hipError_t err = hipSuccess;
    void small(const SmallArgs &a, uint32_t batch) {
        ProfScope ps(ctx, "ntt_small_kernel", (4.0 * a.n_in + 4.0 * (1ull << a.L)) * batch);
        ntt_small_kernel<<<batch, SMI_NTT_THREADS, 0, ctx->stream>>>(a);
        note();
    }
    template <int LR, int LW, int KIND, int CAP> void launch(const PassArgs &a) {
        if constexpr (KIND != PASS_FIRST) {
            if (ctx->ntt_share_cols && NttPass<LR, LW, KIND, CAP>::share_cols(a)) {
                const dim3 grid(a.n_tiles, NttPass<LR, LW, KIND, CAP>::col_groups(a));
                if (KIND == PASS_MID && (a.flags & NTT_TW_IN) && ctx->ntt_twin_regs)
                    ntt_pass_cols_kernel<LR, LW, KIND, CAP, KIND == PASS_MID><<<grid, 1 << (LR + LW - 4), 0, ctx->stream>>>(a);
                else
                    ntt_pass_cols_kernel<LR, LW, KIND, CAP><<<grid, 1 << (LR + LW - 4), 0, ctx->stream>>>(a);
                return;
            }
        }
        ntt_pass_kernel<LR, LW, KIND, CAP><<<dim3(a.n_tiles, a.batch), 1 << (LR + LW - 4), 0, ctx->stream>>>(a);
    }
    template <int LR, int LW> void launch_probe(int kind, const PassArgs &a0) {
        static const uint32_t lin = getenv("SMI_PROBE_LINEAR") ? (uint32_t)atoi(getenv("SMI_PROBE_LINEAR")) & 3u : 0u;
        PassArgs a = a0;
        a.flags |= lin << 30;      // the two top bits of flags are free (ntt_core.h)
        const dim3 grid(a.n_tiles, a.batch);
        if (kind == PASS_FIRST) ntt_copy_probe_kernel<LR, LW, PASS_FIRST><<<grid, 1 << (LR + LW - 4), 0, ctx->stream>>>(a);
        else if (kind == PASS_MID) ntt_copy_probe_kernel<LR, LW, PASS_MID><<<grid, 1 << (LR + LW - 4), 0, ctx->stream>>>(a);
        else ntt_copy_probe_kernel<LR, LW, PASS_LAST><<<grid, 1 << (LR + LW - 4), 0, ctx->stream>>>(a);
    }
    template <int LR, int LW, int CAP> void launch_kind(int kind, const PassArgs &a) {
        if (kind == PASS_FIRST) launch<LR, LW, PASS_FIRST, CAP>(a);
        else if (kind == PASS_MID) launch<LR, LW, PASS_MID, CAP>(a);
        else launch<LR, LW, PASS_LAST, CAP>(a);
    }
    void pass(int logr, int logw, bool last, const PassArgs &a, uint32_t batch) {
        // algorithmic bytes of one pass: every point read once and written once (4 B each);
        // the first pass of a zero-padded transform reads only its n_in real inputs
        const double n = (double)(1ull << (a.L - a.shard_log));   // points this launch moves (a strip of a sharded transform)
        const double bytes = ((a.flags & NTT_FIRST) && !a.shard_log ? 4.0 * a.n_in : 4.0 * n) * batch + 4.0 * n * batch;
        const bool wide = a.F.p < (1u << 29);   // lazy range 8p instead of 4p (ntt_core.h)
        const int kind = last ? PASS_LAST : (a.flags & NTT_FIRST) ? PASS_FIRST : PASS_MID;
#define X(LR, LW)                                                                                              \
    if (logr == LR && logw == LW) {                                                                            \
        if (ctx->copy_probe) {                                                                                 \
            ProfScope ps(ctx, kind == PASS_LAST ? "ntt_copy_probe<" #LR "," #LW ",last>" : kind == PASS_MID ? "ntt_copy_probe<" #LR "," #LW ",mid>" : "ntt_copy_probe<" #LR "," #LW ",first>", bytes); \
            launch_probe<LR, LW>(kind, a);                                                                     \
            note();                                                                                            \
            return;                                                                                            \
        }                                                                                                      \
        const bool cols = ctx->ntt_share_cols && (kind == PASS_LAST ? NttPass<LR, LW, PASS_LAST, 4>::share_cols(a)                     \
                                                  : kind == PASS_MID ? NttPass<LR, LW, PASS_MID, 4>::share_cols(a) : false);            \
        ProfScope ps(ctx, cols ? (kind == PASS_LAST ? "ntt_pass_cols_kernel<" #LR "," #LW ",last>" : "ntt_pass_cols_kernel<" #LR "," #LW ",mid>") \
                          : kind == PASS_LAST ? "ntt_pass_kernel<" #LR "," #LW ",last>" : kind == PASS_MID ? "ntt_pass_kernel<" #LR "," #LW ",mid>" : "ntt_pass_kernel<" #LR "," #LW ",first>", bytes); \
        if (wide) launch_kind<LR, LW, 8>(kind, a);                                                             \
        else launch_kind<LR, LW, 4>(kind, a);                                                                  \
        note();                                                                                                \
        return;                                                                                                \
    }
        SMI_NTT_SHAPES(X)
#undef X
        if (err == hipSuccess) err = hipErrorInvalidValue;
    }
    void note() {
        hipError_t e = hipGetLastError();
        if (err == hipSuccess) err = e;
    }
};
}  // namespace

// Extension of `batch` coefficient columns (n = 2^log_n each, already scaled by the coset offset's
// powers) to N = n << log_blowup evaluations in two passes; d_coef may be the head of d_out's columns.
int dev_lde2(smi_ctx *ctx, const uint32_t *d_coef, uint32_t *d_out, uint32_t log_n, uint32_t log_blowup, uint32_t batch,
             size_t coef_stride, size_t out_stride) {
    if (!lde2_supported(log_n, log_blowup) || log_n + log_blowup > ctx->fs.K) return smi_fail(ctx, SMI_ERR_BAD_ARG, "lde2: unsupported shape");
    if (!batch) return SMI_OK;
    if (batch > 65535) return smi_fail(ctx, SMI_ERR_BAD_ARG, "batch too large");
    const uint32_t logN = log_n + log_blowup;
    LdeArgs a;
    memset(&a, 0, sizeof a);
    a.coef = d_coef; a.out = d_out; a.coef_stride = coef_stride; a.out_stride = out_stride;
    a.F = ctx->fs.F; a.T = ctx_tables(ctx, 0); a.L = log_n; a.beta = log_blowup; a.batch = batch;
    SMI_TRY(ctx_root_table(ctx, log_n - SMI_LDE_LOGB + log_blowup, &a.ctab));
    SMI_TRY(ctx_scratch(ctx, ((size_t)batch << logN) + ((size_t)batch << log_n), &a.mid));
    a.coef_t = a.mid + ((size_t)batch << logN);
    const bool wide = a.F.p < (1u << 29);
    const double n = (double)(1ull << log_n), N = (double)(1ull << logN);
    static const uint32_t dbg = getenv("SMI_LDE_DBG") ? (uint32_t)atoi(getenv("SMI_LDE_DBG")) : 0u;   // tuning runs only
    static const int env_geo = getenv("SMI_LDE_GEO") ? atoi(getenv("SMI_LDE_GEO")) : -1, env_lay = getenv("SMI_LDE_LAYOUT") ? atoi(getenv("SMI_LDE_LAYOUT")) : -1;
    a.dbg = dbg;
    a.geo_rq = env_geo >= 0 && (uint32_t)env_geo <= log_blowup && env_geo <= 4 ? (uint32_t)env_geo : lde_default_geo_rq(log_blowup);
    a.lay_kq = env_lay >= (int)(SMI_LDE_BLINES_LOG - a.geo_rq) && env_lay <= 4 ? (uint32_t)env_lay : SMI_LDE_BLINES_LOG - a.geo_rq;
    const int logr = (int)log_n - SMI_LDE_LOGB;
    {
        ProfScope ps(ctx, "lde_coef_tile_kernel", 8.0 * n * batch);
        lde_coef_tile_kernel<<<dim3(256 / LdeCoefTile::T, (1u << logr) / LdeCoefTile::T, batch), LdeCoefTile::NT, 0, ctx->stream>>>(a);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (ctx->copy_probe) {   // the copy-only twins of both passes (bench.py's pattern-copy figure)
        a.n_tiles = 256u << log_blowup;
        {
            ProfScope ps(ctx, logr == 12 ? "lde_copy_probe_a<12>" : logr == 11 ? "lde_copy_probe_a<11>" : "lde_copy_probe_a<10>", (4.0 * n + 4.0 * N) * batch);
            const dim3 grid(a.n_tiles, batch);
            if (logr == 10) lde_a_probe_kernel<10><<<grid, 256, 0, ctx->stream>>>(a);
            if (logr == 11) lde_a_probe_kernel<11><<<grid, 512, 0, ctx->stream>>>(a);
            if (logr == 12) lde_a_probe_kernel<12><<<grid, 1024, 0, ctx->stream>>>(a);
        }
        a.n_tiles = 1u << (logN - SMI_LDE_LOGB - SMI_LDE_BLINES_LOG);
        {
            ProfScope ps(ctx, "lde_copy_probe_b", 8.0 * N * batch);
            lde_b_probe_kernel<<<dim3(((a.dbg >> 8) & 255u) == 3 ? a.n_tiles / 2 : a.n_tiles, batch), 1024, 0, ctx->stream>>>(a);
        }
        HIP_TRY(ctx, hipGetLastError());
        return SMI_OK;
    }
    {
        a.n_tiles = 256u << log_blowup;
        const dim3 grid(a.n_tiles, batch);
        ProfScope ps(ctx, logr == 12 ? "lde_a_kernel<12>" : logr == 11 ? "lde_a_kernel<11>" : "lde_a_kernel<10>", (4.0 * n + 4.0 * N) * batch);
#define LA(LR)                                                                                     \
    if (logr == LR) {                                                                              \
        if (wide) lde_a_kernel<LR, 8><<<grid, 1 << (LR - 2), 0, ctx->stream>>>(a);                 \
        else lde_a_kernel<LR, 4><<<grid, 1 << (LR - 2), 0, ctx->stream>>>(a);                      \
    }
        LA(10) LA(11) LA(12)
#undef LA
        HIP_TRY(ctx, hipGetLastError());
    }
    {
        a.n_tiles = 1u << (logN - SMI_LDE_LOGB - SMI_LDE_BLINES_LOG);
        const dim3 grid(a.n_tiles, batch);
        ProfScope ps(ctx, "lde_b_kernel", 8.0 * N * batch);
        if (wide) lde_b_kernel<8><<<grid, 1024, 0, ctx->stream>>>(a);
        else lde_b_kernel<4><<<grid, 1024, 0, ctx->stream>>>(a);
        HIP_TRY(ctx, hipGetLastError());
    }
    return SMI_OK;
}

int dev_ntt(smi_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint32_t log_n, size_t n_in, uint32_t batch,
            size_t in_stride, size_t out_stride, int inverse, uint64_t offset, uint64_t post_scale) {
    const uint32_t p = ctx->fs.F.p;
    if (log_n > ctx->fs.K)
        return smi_fail(ctx, p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME,
                        "transform size exceeds the two-adicity of the modulus");
    const uint64_t n = 1ull << log_n;
    if (!batch) return SMI_OK;
    if (n_in > n || (inverse && n_in != n)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "n_in out of range");
    if (offset >= p || post_scale >= p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "offset/post_scale must be < p");
    if (offset == 0) return smi_fail(ctx, SMI_ERR_NO_INVERSE, "offset must be invertible");
    if (batch > 65535) return smi_fail(ctx, SMI_ERR_BAD_ARG, "batch too large");
    // d_in == d_out is always safe: every pass (and the small kernel) has read all of its
    // input into LDS / the scratch buffer before anything is written to d_out.

    if (n_in == 0) {  // Polynomial::eval_domain of the empty polynomial: all zeros (src/univariate/eval.rs:6-14)
        for (uint32_t c = 0; c < batch; c++) HIP_TRY(ctx, hipMemsetAsync(d_out + (size_t)c * out_stride, 0, n * 4, ctx->stream));
        return SMI_OK;
    }
    ScaleScope pin__(ctx);   // rq.S is read by the launches below
    NttRequest rq;
    memset(&rq, 0, sizeof rq);
    rq.in = d_in; rq.out = d_out; rq.L = log_n; rq.n_in = (uint32_t)n_in; rq.batch = batch;
    rq.in_stride = in_stride; rq.out_stride = out_stride; rq.F = ctx->fs.F;
    rq.T = ctx_tables(ctx, inverse);
    if (!inverse) {
        rq.pre_scale = offset != 1;
        rq.q_plain = (uint32_t)offset;
        if (rq.pre_scale) SMI_TRY(ctx_scale_tables(ctx, 1, (uint32_t)offset, log_n, &rq.S));
    } else {
        rq.post_scale = true;  // n^-1 * (post_scale/offset)^j
        const uint32_t ninv = h_inv(ctx, (uint32_t)(n % p));
        const uint32_t q = h_mul(ctx, (uint32_t)post_scale, h_inv(ctx, (uint32_t)offset));
        rq.q_plain = q;
        SMI_TRY(ctx_scale_tables(ctx, ninv, q, log_n, &rq.S));
    }
    rq.defer_tw = ctx->ntt_defer_tw;
    rq.last_direct = ctx->ntt_last_direct;
    if (ntt_make_plan(log_n, batch).np > 0) SMI_TRY(ctx_scratch(ctx, (size_t)batch << log_n, &rq.scratch));   // inter-pass buffer
    HipLauncher ln{ctx};
    if (!ntt_run(ln, rq)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "ntt: multi-pass plan without its inter-pass buffer");
    if (ln.err != hipSuccess) return smi_hip_fail(ctx, ln.err, "ntt kernel launch");
    return SMI_OK;
}

// One transform sharded over 2^log_g ranks on the pass pipeline (ntt_driver.h): pass 0 on this rank's
// strip in place, and -- after the caller's exchange -- the remaining passes.
static int shard_request(smi_ctx *ctx, uint32_t log_n, int inverse, uint64_t offset, NttRequest *rq) {
    const uint32_t p = ctx->fs.F.p;
    if (log_n > ctx->fs.K)
        return smi_fail(ctx, p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "transform size exceeds the two-adicity of the modulus");
    if (offset >= p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "offset must be < p");
    if (offset == 0) return smi_fail(ctx, SMI_ERR_NO_INVERSE, "offset must be invertible");
    if (inverse && offset != 1) return smi_fail(ctx, SMI_ERR_BAD_ARG, "sharded inverse transform: offset must be 1");
    memset(rq, 0, sizeof *rq);
    rq->L = log_n; rq->n_in = (uint32_t)(1ull << log_n); rq->batch = 1; rq->F = ctx->fs.F;
    rq->T = ctx_tables(ctx, inverse);
    if (!inverse) {
        rq->pre_scale = offset != 1;
        rq->q_plain = (uint32_t)offset;
        if (rq->pre_scale) SMI_TRY(ctx_scale_tables(ctx, 1, (uint32_t)offset, log_n, &rq->S));
    } else {
        rq->post_scale = true;   // n^-1, constant
        rq->q_plain = 1;
        SMI_TRY(ctx_scale_tables(ctx, h_inv(ctx, (uint32_t)((1ull << log_n) % p)), 1, log_n, &rq->S));
    }
    return SMI_OK;
}
int dev_ntt_shard_first(smi_ctx *ctx, uint32_t *d_strip, uint32_t log_n, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset) {
    ScaleScope pin__(ctx);
    NttRequest rq;
    SMI_TRY(shard_request(ctx, log_n, inverse, offset, &rq));
    rq.in = d_strip; rq.out = d_strip;
    HipLauncher ln{ctx};
    if (!ntt_run_shard_first(ln, rq, log_g, rank)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "sharded transform: size too small for this many ranks");
    if (ln.err != hipSuccess) return smi_hip_fail(ctx, ln.err, "ntt kernel launch");
    return SMI_OK;
}
int dev_ntt_shard_rest(smi_ctx *ctx, uint32_t *d_rows, uint32_t *d_out, uint32_t log_n, uint32_t log_g, int inverse) {
    ScaleScope pin__(ctx);
    NttRequest rq;
    SMI_TRY(shard_request(ctx, log_n, inverse, 1, &rq));
    rq.scratch = d_rows; rq.out = d_out;
    rq.last_direct = ctx->ntt_last_direct;
    HipLauncher ln{ctx};
    if (!ntt_run_shard_rest(ln, rq, log_g)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "sharded transform: size too small for this many ranks");
    if (ln.err != hipSuccess) return smi_hip_fail(ctx, ln.err, "ntt kernel launch");
    return SMI_OK;
}
