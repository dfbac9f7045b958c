// lde_core.h -- the per-thread phases of the two-pass low-degree extension (host+device).
//
// Replaces, for a batch of trace columns, Polynomial::eval_domain over the blowup coset
// (reference src/univariate/eval.rs:16-21) of the polynomial interpolate_domain produced
// (src/univariate/interpolate.rs:6-44).  The generic transform (ntt_core.h) needs three passes
// over the 2^(L+beta) outputs; here the zero padding of the extension is used to do it in two.
//
// With N = n * 2^beta, Omega a primitive N-th root and k = r + 2^beta * q (r: coset, q < n):
//     E[r + 2^beta q] = sum_j (c[j] Omega^(r j)) w_n^(j q)
// i.e. the extension is 2^beta coset transforms of the SAME n coefficients, interleaved.  Split
// n = R * 2^10, j = j1 * 2^10 + j0, q = k1 + R * k0:
//   pass A (per coset r, tile = 4 adjacent j0 x all R j1):
//     Y_r[k1][j0] = Omega^(j0 (r + 2^beta k1)) * sum_j1 (c[j1 2^10 + j0] Omega^(r 2^10 j1)) w_R^(j1 k1)
//   pass B (tile = 1024 j0 x 16 lines (k1, r)):
//     E[r + 2^beta (k1 + R k0)] = sum_j0 Y_r[k1][j0] w_1024^(j0 k0)
// Pass A reads n and writes N elements, pass B reads and writes N: (n + 3N) * 4 bytes per column
// against (n + 5N) * 4 for three passes.
//
// Both private buffers are laid out for the side that is bound by memory:
//   * the coefficients are first regrouped (lde_coef_tile: a transposition of 16-byte elements, n
//     elements in and out) into [j0 >> 2][j1][j0 & 3], so that the input tile of pass A -- four
//     adjacent j0, every j1 -- is one contiguous 16 R-byte run instead of R 16-byte pieces of
//     128-byte lines (measured: those pieces cost 8x their bytes between L2 and L1, and every tile is
//     read once per coset);
//   * the intermediate Y is [k1 >> kq][r][j0 >> 2][k1 & (2^kq - 1)][j0 & 3], kq = log2 of the adjacent k1
//     in a tile of pass B (16 lines = 2^rq cosets x 2^kq k1): a B tile reads 2^rq contiguous runs of
//     2^(kq+12) bytes.  A's last in-tile step is mapped so that a wave stores 16 consecutive k1 x 4 j0,
//     i.e. 2^(4-kq) pieces of 2^(kq+4) bytes (pass A is bound by arithmetic, its stores are fire and
//     forget).  B writes 2^beta-interleaved natural order: 64 bytes of every 128-byte output line per
//     tile; the tile holding the other half is its neighbour in tile order and therefore runs on the
//     same XCD at about the same time (xcd_tile).
#pragma once
#include "ntt_core.h"

template <> struct Steps<12> { enum { n = 3, s0 = 4, s1 = 4, s2 = 4 }; };

#define SMI_LDE_LOGB 10                  // pass B transforms 1024-point lines
#define SMI_LDE_BLINES_LOG 4             // 16 lines per B tile

struct LdeArgs {
    const uint32_t *coef;   // column c: n = 2^L coefficients at coef + c * coef_stride (natural order)
    uint32_t *coef_t;       // batch * 2^L elements: the same coefficients as [j0 >> 2][j1][j0 & 3], column c at c << L
    uint32_t *mid;          // batch * 2^(L+beta) elements (layout above), column c at c << (L+beta)
    uint32_t *out;          // column c: N evaluations at out + c * out_stride (natural order)
    uint64_t coef_stride, out_stride;
    Fp F;
    NttTables T;            // forward direction
    const Tw2 *ctab;        // w_M^e, e < M = 2^(L - 10 + beta), with Shoup quotients: Omega^(2^10 r j1) = ctab[(r j1) mod M]
    uint32_t L, beta;
    uint32_t n_tiles;       // grid.x of the launch this struct goes to
    uint32_t batch;         // grid.y
    uint32_t lay_kq;        // the intermediate keeps 2^lay_kq adjacent k1 together: [k1 >> lay_kq][r][j0 >> 2][k1 & (2^lay_kq - 1)][j0 & 3]
    uint32_t geo_rq;        // a tile of pass B holds 2^geo_rq cosets x 2^(4 - geo_rq) adjacent k1 (geo_rq <= beta, 4 - geo_rq <= lay_kq)
    uint32_t dbg;           // tuning runs only (SMI_LDE_DBG): 1 = pass A computes but does not store, 2 = same for pass B
};

// LDS row swizzle of pass A's tile (4 words per row, no padding): XOR the two upper nibbles of the
// row index into the lowest one.  A bijection on rows whatever LOGR is; for every access pattern
// of the tile program the 16 rows a wave touches differ in exactly one nibble, so their low
// nibbles stay distinct and a wave's 64 words fall into 64 different banks.
// A tile of pass B holds 16 lines = 2^rq cosets x 2^kq adjacent k1, rq = min(beta, 2)
SMI_HD uint32_t lde_kq_bits(uint32_t beta) { return SMI_LDE_BLINES_LOG - (beta < 2 ? beta : 2u); }

// It is linear over GF(2): for a base and an offset with no common bits, swz(base + offset) =
// swz(base) ^ swz(offset), and the offsets below are compile-time constants after unrolling -- one XOR
// per LDS access instead of the whole expression.
SMI_HD constexpr uint32_t lde_swz(uint32_t row) { return row ^ ((row >> 4) & 15u) ^ ((row >> 8) & 15u); }

// lde_coef_tile: coefficients [j1][j0] -> [j0 >> 2][j1][j0 & 3], as a transposition of the matrix of
// 16-byte elements [R rows (j1)][256 columns (jt)].  One workgroup (256 threads) moves a 32 x 32 tile
// through LDS: reads 512-byte runs, writes 512-byte runs.
struct LdeCoefTile {
    enum { T = 32, NT = 256 };
    // phase 1: global -> LDS; phase 2: LDS -> global.  tile: T * (T + 1) elements of 4 words
    static SMI_HD void load(const LdeArgs &a, uint32_t bx, uint32_t by, uint32_t batch, uint32_t *tile, uint32_t tid) {
        const uint32_t *col = a.coef + (uint64_t)batch * a.coef_stride;
#pragma unroll
        for (int i = 0; i < T * T / NT; i++) {
            const uint32_t e = tid + i * NT, row = e >> 5, c = e & 31u;      // row: j1 within the tile, c: jt within the tile
            const uint32_t j1 = by * T + row, jt = bx * T + c;
#pragma unroll
            for (int w = 0; w < 4; w++) tile[(row * (T + 1) + c) * 4 + w] = ld32(col, (j1 << SMI_LDE_LOGB) + (jt << 2) + w);
        }
    }
    static SMI_HD void store(const LdeArgs &a, uint32_t bx, uint32_t by, uint32_t batch, const uint32_t *tile, uint32_t tid) {
        uint32_t *dst = a.coef_t + ((uint64_t)batch << a.L);
        const uint32_t logR = a.L - SMI_LDE_LOGB;
#pragma unroll
        for (int i = 0; i < T * T / NT; i++) {
            const uint32_t e = tid + i * NT, c = e >> 5, row = e & 31u;
            const uint32_t j1 = by * T + row, jt = bx * T + c;
#pragma unroll
            for (int w = 0; w < 4; w++) st32(dst, (((jt << logR) + j1) << 2) + w, tile[(row * (T + 1) + c) * 4 + w]);
        }
    }
};

template <int LOGR, int CAP> struct LdeA {
    typedef Steps<LOGR> St;
    enum { R = 1 << LOGR, LOGW = 2, W = 4, TILE = R * W, NT = TILE / 16, V = 16, RL = 1 << St::s2, NB = 16 / RL };
    static_assert(St::n == 3 && St::s0 == 4, "three-step lines, radix-16 first step");

    struct TileId {
        uint32_t r;    // coset
        uint32_t jt;   // j0 >> 2
    };
    static SMI_HD TileId tile_id(const LdeArgs &a, uint32_t block) {
        const uint32_t t = xcd_tile(block, a.n_tiles);
        TileId id;
        id.r = t & ((1u << a.beta) - 1u);     // the 2^beta cosets of one input tile are neighbours: one HBM read
        id.jt = t >> a.beta;
        return id;
    }
    // LDS holds only w_{R/16}^j (what the middle and last steps and every radix-16/8 butterfly use): the
    // first step's twiddles w_R^(pos*kk) come straight from the context's table in global memory (L2
    // resident, 15 reads per thread).  With the whole w_R table staged the tile would leave room for
    // one workgroup per CU; this way two fit (66 KB each at R = 4096).
    enum { TWS = R / 16 };
    static SMI_HD void load_tw(const LdeArgs &a, Tw2 *tws, uint32_t tid) {
        if (tid < (uint32_t)TWS) tws[tid] = a.T.tw10[tid << (SMI_TW_LOG + 4 - LOGR)];
    }
    // rows pos + i*R/16 of column w: the inputs of the thread's radix-16 butterfly, times Omega^(r 2^10 j1)
    static SMI_HD void load_regs(const LdeArgs &a, const TileId &t, uint32_t batch, uint32_t (&v)[V], uint32_t tid) {
        const uint32_t pos = tid >> 2;
        const uint32_t *col = a.coef_t + ((uint64_t)batch << a.L);
        const uint32_t o0 = (t.jt << (LOGR + 2)) + tid;       // [jt][j1 = pos + i R/16][w]: a wave reads 256 contiguous bytes
#pragma unroll
        for (int i = 0; i < V; i++) v[i] = ld32(col, o0 + (uint32_t)(i * (R / 4)));
        if (t.r) {
            // the coset's input scale Omega^(r 2^10 j1) = w_M^(r j1): one table read and one Shoup product per
            // input (a running product costs 13 instructions per input against these 4 + a load; the
            // generic first pass pays 14 products per 16 outputs for its three degenerate stages)
            const uint32_t mask = (1u << (LOGR + a.beta)) - 1u;
            const uint32_t e0 = t.r * pos, es = t.r * (uint32_t)(R / 16);
#pragma unroll
            for (int i = 0; i < V; i++) v[i] = shoup_mul(v[i], ld_tw(a.ctab, (e0 + (uint32_t)i * es) & mask), a.F.p);   // [0, 2p)
        }
    }
    static SMI_HD void step0(const LdeArgs &a, uint32_t (&x)[V], uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        const uint32_t w = tid & 3u, pos = tid >> 2, pos_sw = lde_swz(pos);
        int m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = 2;      // scaled inputs are Shoup products in [0, 2p) (coset 0: canonical)
        Tw2 t0[16];
#pragma unroll
        for (int kk = 1; kk < 16; kk++) t0[kk] = ld_tw(a.T.tw10, ((pos * kk) & (R - 1)) << (SMI_TW_LOG - LOGR));
        dft_regs<4, CAP>(x, m, tw, LOGR - 8, a.F);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            uint32_t v = x[brev<4>(kk)];
            int mv = m[brev<4>(kk)];
            if (kk) v = shoup_mul(v, t0[kk], a.F.p);
            else lz_fold_to2(v, mv, a.F.p);
            tile[((pos_sw ^ lde_swz((uint32_t)kk << (LOGR - 4))) << 2) + w] = v;   // row pos + kk R/16 (pos < R/16)
        }
    }
    static SMI_HD void step_mid(const LdeArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        enum { S = St::s1, r = 1 << S, MLOG = LOGR - 4, SUB = MLOG - S, NBM = (TILE / r) / NT };
#pragma unroll
        for (int bi = 0; bi < NBM; bi++) {
            const uint32_t u = tid + bi * NT;
            const uint32_t w = u & 3u, ub = u >> 2;
            const uint32_t blk = ub >> SUB, pos = ub & ((1u << SUB) - 1u);
            const uint32_t base_sw = lde_swz((blk << MLOG) + pos);     // pos < 2^SUB: the q << SUB below share no bits with it
            uint32_t x[r];
            int m[r];
#pragma unroll
            for (int q = 0; q < r; q++) {
                x[q] = tile[((base_sw ^ lde_swz((uint32_t)q << SUB)) << 2) + w];
                m[q] = 2;
            }
            dft_regs<S, CAP>(x, m, tw, MLOG - S, a.F);
#pragma unroll
            for (int kk = 0; kk < r; kk++) {
                uint32_t v = x[brev<S>(kk)];
                int mv = m[brev<S>(kk)];
                if (kk) v = shoup_mul(v, tw[(pos * kk) & (TWS - 1)], a.F.p);   // w_R^(16 pos kk)
                else lz_fold_to2(v, mv, a.F.p);
                tile[((base_sw ^ lde_swz((uint32_t)kk << SUB)) << 2) + w] = v;
            }
        }
    }
    // Last in-tile step fused with the inter-pass twiddle and the store.  Thread -> butterfly
    // mapping: w = lane & 3, d0 (first step's digit = k1 bits 0..3) = next four lane bits, d1 (middle
    // digit) = wave index: a wave's store at a fixed kk is 16 consecutive k1 x 4 j0 = 256 contiguous
    // bytes of the intermediate.
    static SMI_HD void last_step_store(const LdeArgs &a, const TileId &t, uint32_t batch, const uint32_t *tile, const Tw2 *tw,
                                       uint32_t tid) {
        enum { S1 = St::s1, SL = St::s2 };
        uint32_t *mid = a.mid + ((uint64_t)batch << (a.L + a.beta));
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const uint32_t u = tid + bi * NT;
            const uint32_t w = u & 3u, d0 = (u >> 2) & 15u, d1 = u >> 6;
            const uint32_t blk_sw = lde_swz(((d0 << S1) | d1) * RL);
            uint32_t x[RL];
            int m[RL];
#pragma unroll
            for (int q = 0; q < RL; q++) {
                x[q] = tile[((blk_sw ^ lde_swz((uint32_t)q)) << 2) + w];
                m[q] = 2;
            }
            dft_regs<SL, CAP>(x, m, tw, LOGR - 4 - SL, a.F);
            // (the inter-pass twiddle Omega^(j0 (r + 2^beta k1)) is applied by pass B as it reads: B is bound by
            // memory and has the issue slots, this pass is bound by arithmetic)
            const uint32_t k1b = d0 | (d1 << 4);
            // [k1 >> kq][r][jt][k1 & (2^kq - 1)][w]; kk only moves k1 >> kq (kq <= 4 + S1)
            const uint32_t kq = a.lay_kq;
            const uint32_t o0 = ((((k1b >> kq) << a.beta) + t.r) << (10 + kq)) + (t.jt << (kq + 2)) + ((k1b & ((1u << kq) - 1u)) << 2) + w;
#pragma unroll
            for (int kk = 0; kk < RL; kk++) {
                const uint32_t val = lz_canon_m(x[brev<SL>(kk)], m[brev<SL>(kk)], a.F.p);
                if (!((a.dbg & 1u) && val != 0xFFFFFFFFu)) st32(mid, o0 + ((uint32_t)kk << (14 + S1 + a.beta)), val);
            }
        }
    }
};

// Copy-only twins (smi_ctx_copy_probe): the same loads and the same store addresses, no arithmetic.
template <int LOGR> struct LdeAProbe {
    typedef LdeA<LOGR, 4> A;
    static SMI_HD void run(const LdeArgs &a, uint32_t block, uint32_t batch, uint32_t tid) {
        typedef Steps<LOGR> St;
        const typename A::TileId t = A::tile_id(a, block);
        uint32_t v[16];
        LdeArgs b = a;
        const typename A::TileId t0 = {0u, t.jt};          // the loads without the coset scale
        A::load_regs(b, t0, batch, v, tid);
        uint32_t *mid = a.mid + ((uint64_t)batch << (a.L + a.beta));
#pragma unroll
        for (int bi = 0; bi < A::NB; bi++) {
            const uint32_t u = tid + bi * A::NT;
            const uint32_t w = u & 3u, d0 = (u >> 2) & 15u, d1 = u >> 6, k1b = d0 | (d1 << 4);
            // tuning knob (SMI_LDE_DBG bits 16..23): store pieces of 2^(kq+4) bytes, kq = knob - 1 (0: the kernel's own)
            const uint32_t knob = (a.dbg >> 16) & 255u, kq = knob ? knob - 1u : a.lay_kq;
            const uint32_t o0 = ((((k1b >> kq) << a.beta) + t.r) << (10 + kq)) + (t.jt << (kq + 2)) + ((k1b & ((1u << kq) - 1u)) << 2) + w;
#pragma unroll
            for (int kk = 0; kk < A::RL; kk++) st32(mid, o0 + ((uint32_t)kk << (14 + St::s1 + a.beta)), v[bi * A::RL + kk] + 1u);
        }
    }
};

// Pass B: the in-tile transform is the generic last pass's (1024-point lines, 16 of them, steps
// 16 x 8 x 8 through LDS); only where the lines come from and where they go differs.
template <int CAP> struct LdeB {
    typedef NttPass<SMI_LDE_LOGB, SMI_LDE_BLINES_LOG, PASS_LAST, CAP> NP;
    enum { R = NP::R, W = NP::W, WP = NP::WP, NT = NP::NT, V = 16, RL = NP::RL, SL = NP::SL, NB = (NP::TILE / NP::RL) / NP::NT };

    struct Geo {   // how the 16 lines split into cosets and adjacent k1 (wave-uniform)
        uint32_t rq_bits, kq_bits, rh_bits;
    };
    static SMI_HD Geo geo(const LdeArgs &a) {
        Geo g;
        g.rq_bits = a.geo_rq;
        g.kq_bits = SMI_LDE_BLINES_LOG - g.rq_bits;
        g.rh_bits = a.beta - g.rq_bits;
        return g;
    }
    struct TileId {
        uint32_t k1_hi, rh;   // k1 >> kq, r >> rq
    };
    static SMI_HD TileId tile_id(const LdeArgs &a, uint32_t block) {
        const Geo g = geo(a);
        const uint32_t t = xcd_tile(block, a.n_tiles);
        TileId id;   // the tiles that share 128-byte output lines are neighbours in tile order
        id.rh = t & ((1u << g.rh_bits) - 1u);
        id.k1_hi = t >> g.rh_bits;
        return id;
    }
    static SMI_HD void load_tw(const LdeArgs &a, Tw2 *tw, uint32_t tid) {
        PassArgs pa;
        pa.T = a.T;
        NP::load_tw(pa, tw, tid);
    }
    // Element e of the tile = (rq, jt, k1q, j0 & 3) in that order; with lay_kq == kq_bits the tile's 16 K inputs
    // are one contiguous run of the intermediate, with a wider layout they are 2^(kq+4)-byte pieces of its
    // 2^(lay_kq+4)-byte blocks (the other pieces belong to the neighbouring tiles).
    static SMI_HD uint32_t in_addr(const LdeArgs &a, const Geo &g, const TileId &t, uint32_t e) {
        const uint32_t rq = e >> (10 + g.kq_bits), jt = (e >> (g.kq_bits + 2)) & 255u, k1q = (e >> 2) & ((1u << g.kq_bits) - 1u), j0lo = e & 3u;
        const uint32_t k1 = (t.k1_hi << g.kq_bits) | k1q, r = (t.rh << g.rq_bits) | rq;
        return ((((k1 >> a.lay_kq) << a.beta) + r) << (10 + a.lay_kq)) + (jt << (a.lay_kq + 2)) + ((k1 & ((1u << a.lay_kq) - 1u)) << 2) + j0lo;
    }
    // With e = i * 1024 + tid, the thread's bits give j0 & 3, k1q and the low 8 - kq bits of jt, the load index i
    // the top kq bits of jt and rq: both addresses are a per-thread base plus wave-uniform multiples of two
    // strides -- one vector add per load instead of re-deriving the fields.
    static SMI_HD void load(const LdeArgs &a, const TileId &t, uint32_t batch, uint32_t (&v)[V], uint32_t tid) {
        static_assert(NT == 1024 && V == 16, "thread / load-index split of the tile's 14 element bits");
        const Geo g = geo(a);
        const uint32_t *mid = a.mid + ((uint64_t)batch << (a.L + a.beta));
        const uint32_t base = in_addr(a, g, t, tid);
        const uint32_t jt_stride = (256u >> g.kq_bits) << (a.lay_kq + 2), r_stride = 1u << (10 + a.lay_kq), imask = (1u << g.kq_bits) - 1u;
#pragma unroll
        for (int i = 0; i < V; i++) v[i] = ld32(mid, base + ((uint32_t)i & imask) * jt_stride + ((uint32_t)i >> g.kq_bits) * r_stride);
    }
    static SMI_HD void to_lds(const LdeArgs &a, const uint32_t (&v)[V], uint32_t *tile, uint32_t tid) {
        const Geo g = geo(a);
        const uint32_t k1q = (tid >> 2) & ((1u << g.kq_bits) - 1u), j0lo = tid & 3u, jt_lo = tid >> (g.kq_bits + 2);
        const uint32_t base = ((jt_lo << 2) + j0lo) * WP + k1q;
        const uint32_t jt_stride = (256u >> g.kq_bits) * 4u * WP, r_stride = 1u << g.kq_bits, imask = (1u << g.kq_bits) - 1u;
#pragma unroll
        for (int i = 0; i < V; i++) tile[base + ((uint32_t)i & imask) * jt_stride + ((uint32_t)i >> g.kq_bits) * r_stride] = v[i];
    }
    // First in-tile step (radix 16 over rows pos + 64 q of line l), with the inter-pass twiddle
    // Omega^(j0 E), E = r + 2^beta k1 of the line, applied to the inputs as they come out of LDS: along a
    // thread's 16 rows it is a geometric sequence, so one running product.
    static SMI_HD void step0(const LdeArgs &a, const TileId &t, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        const Geo g = geo(a);
        const uint32_t l = tid & (W - 1), pos = tid >> SMI_LDE_BLINES_LOG;          // pos < 64
        const uint32_t r = (t.rh << g.rq_bits) | (l >> g.kq_bits), k1 = (t.k1_hi << g.kq_bits) | (l & ((1u << g.kq_bits) - 1u));
        const uint32_t E = r + (k1 << a.beta), sh = a.T.K - (a.L + a.beta);
        uint32_t cur = two_level(a.T.lo, a.T.hi, a.T.h, (pos * E) << sh, a.F);
        const uint32_t ratio = two_level(a.T.lo, a.T.hi, a.T.h, (E << 6) << sh, a.F), rq = ratio * a.F.pinv;
        uint32_t x[16];
        int m[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            x[q] = mont_mul(tile[(pos + ((uint32_t)q << 6)) * WP + l], cur, a.F);
            m[q] = 1;
            if (q + 1 < 16) cur = mont_mul_c(cur, ratio, rq, a.F);
        }
        dft_regs<4, CAP>(x, m, tw, SMI_LDE_LOGB - 4, a.F);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            uint32_t v = x[brev<4>(kk)];
            int mv = m[brev<4>(kk)];
            if (kk) v = shoup_mul(v, tw[(pos * kk) & (R - 1)], a.F.p);
            else lz_fold_to2(v, mv, a.F.p);
            tile[(pos + ((uint32_t)kk << 6)) * WP + l] = v;
        }
    }
    static SMI_HD void step_mid(const LdeArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        PassArgs pa;
        pa.F = a.F;
        NP::step_mid(pa, tile, tw, tid);
    }
    static SMI_HD void last_step_store(const LdeArgs &a, const TileId &t, uint32_t batch, const uint32_t *tile, const Tw2 *tw,
                                       uint32_t tid) {
        const Geo g = geo(a);
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride;
        const uint32_t logRA = a.L - SMI_LDE_LOGB, ksh = a.beta + logRA;   // output index = r + 2^beta (k1 + R_A k0)
        const uint32_t base = ((t.k1_hi << g.kq_bits) << a.beta) + (t.rh << g.rq_bits);
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const uint32_t u = tid + bi * NT;
            const uint32_t l = u & (W - 1), blk = u >> SMI_LDE_BLINES_LOG;
            uint32_t x[RL];
            int m[RL];
#pragma unroll
            for (int q = 0; q < RL; q++) {
                x[q] = tile[(blk * RL + q) * WP + l];
                m[q] = 2;
            }
            dft_regs<SL, CAP>(x, m, tw, SMI_LDE_LOGB - SL, a.F);
            const uint32_t k0b = NP::blk_to_k(blk);
            const uint32_t lineoff = ((l & ((1u << g.kq_bits) - 1u)) << a.beta) + (l >> g.kq_bits);
            const uint32_t o0 = (k0b << ksh) + base + lineoff;
#pragma unroll
            for (int kk = 0; kk < RL; kk++) {
                const uint32_t val = lz_canon_m(x[brev<SL>(kk)], m[brev<SL>(kk)], a.F.p);
                if (!((a.dbg & 2u) && val != 0xFFFFFFFFu)) st32(out, o0 + ((uint32_t)kk << (SMI_LDE_LOGB - SL + ksh)), val);
            }
        }
    }
};

struct LdeBProbe {
    typedef LdeB<4> B;
    // Tuning knob (SMI_LDE_DBG bits 8..15, copy-only runs): alternative access patterns of pass B, to
    // price them without building the arithmetic around them.
    //   0 the kernel's own pattern                 1 lines = 2 k1 x 8 cosets (64 contiguous bytes per output line)
    //   2 lines = 8 k1 x 2 cosets                   3 both halves of every output line by ONE workgroup, one after the other
    //   4 the kernel's pattern without xcd_tile     5 whole 128-byte output lines (half as many rows, 8 bytes per lane)
    static SMI_HD void run(const LdeArgs &a, uint32_t block, uint32_t batch, uint32_t tid) {
        const uint32_t var = (a.dbg >> 8) & 255u;
        B::Geo g = B::geo(a);
        if (var == 1 && a.beta >= 3) { g.rq_bits = 3; g.kq_bits = 1; g.rh_bits = a.beta - 3; }
        if (var == 2) { g.rq_bits = 1; g.kq_bits = 3; g.rh_bits = a.beta - 1; }
        const uint32_t *mid = a.mid + ((uint64_t)batch << (a.L + a.beta));
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride;
        const uint32_t ksh = a.beta + a.L - SMI_LDE_LOGB;
        const uint32_t reps = var == 3 ? 2u : 1u;
        for (uint32_t rep = 0; rep < reps; rep++) {
            uint32_t t = var == 4 ? block : xcd_tile(block, a.n_tiles);
            if (var == 3) t = 2u * xcd_tile(block, a.n_tiles / 2u) + rep;     // launched with half as many workgroups
            B::TileId id;
            id.rh = t & ((1u << g.rh_bits) - 1u);
            id.k1_hi = t >> g.rh_bits;
            uint32_t v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = ld32(mid, B::in_addr(a, g, id, (uint32_t)(i * B::NT) + tid));
            const uint32_t base = ((id.k1_hi << g.kq_bits) << a.beta) + (id.rh << g.rq_bits);
            if (var == 5) {   // lane pairs write 8 bytes each: 16 lanes cover one whole 128-byte line of one row
                const uint32_t l = tid & 15u, row0 = tid >> 4;
                const uint32_t line0 = ((id.k1_hi << g.kq_bits) << a.beta) & ~31u;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t k0 = row0 + (uint32_t)i * 64u + ((id.rh & 1u) << 9);   // the partner tile takes the other 512 rows
                    st32(out, (k0 << ksh) + line0 + 2u * l, v[2 * i] + 1u);
                    st32(out, (k0 << ksh) + line0 + 2u * l + 1u, v[2 * i + 1] + 1u);
                }
                continue;
            }
#pragma unroll
            for (int bi = 0; bi < B::NB; bi++) {
                const uint32_t u = tid + bi * B::NT;
                const uint32_t l = u & (B::W - 1), blk = u >> SMI_LDE_BLINES_LOG;
                const uint32_t lineoff = ((l & ((1u << g.kq_bits) - 1u)) << a.beta) + (l >> g.kq_bits);
                const uint32_t o0 = (B::NP::blk_to_k(blk) << ksh) + base + lineoff;
#pragma unroll
                for (int kk = 0; kk < B::RL; kk++) st32(out, o0 + ((uint32_t)kk << (SMI_LDE_LOGB - B::SL + ksh)), v[bi * B::RL + kk] + 1u);
            }
        }
    }
};

// Defaults of the layout / tile geometry (overridden by SMI_LDE_LAYOUT = lay_kq, SMI_LDE_GEO = geo_rq in tuning
// runs): measured in DESIGN.md section 3.
SMI_HD uint32_t lde_default_geo_rq(uint32_t beta) { return beta < 2 ? beta : 2u; }
// Which (log n, log blowup) the two-pass extension serves: lines of pass A are 2^(L-10) points
// (1024, 2048 or 4096), at least two cosets, and the tile counts are multiples of 8 (XCD order).
inline bool lde2_supported(uint32_t L, uint32_t beta) { return L >= 20 && L <= 22 && beta >= 1 && beta <= 4; }
