// ntt_core.h -- the per-thread phases of the NTT kernels (host+device).
//
// MI355X-native replacement for the reference's O(n^3) Lagrange interpolation
// (src/univariate/interpolate.rs:6-44) and O(N*d) power-sum evaluation
// (src/univariate/eval.rs:6-21) on geometric domains offset*omega^k -- the reference
// itself contains no NTT.  Natural order in, natural order out.
//
// Decomposition (in-place decimation in frequency by digits, transposition fused into the
// last pass): n = 2^L = R_0 * R_1 * ... * R_{np-1}.  Pass p views the column as
// [A = prod_{q<p} R_q][R_p][B = n / (A*R_p)] and transforms along the middle axis:
//     Y[a][k][b] = w_m^(k*b) * sum_j X[a][j][b] * w_R^(j*k),     m = R_p * B
// so after the last pass (B = 1) element (k_0,...,k_{np-1}) holds output index
// K = k_0 + R_0*k_1 + R_0*R_1*k_2 + ...; the last pass writes it there directly.
//
// One workgroup = one tile of W lines x R points (4K..16K points), 16 points per thread.
// Lines are W adjacent b-columns (runs of W*4 contiguous bytes in HBM) for strided passes,
// and W rows with adjacent k_0 for the last pass (so its transposed writes are W-element
// runs too).  The R-point transform is 2-3 register-resident steps (radix 16, then 8/4/16)
// with an LDS exchange between them.  Strided passes feed the first step straight from
// their global loads and every pass stores straight from the last step's registers, so a
// two-step tile crosses LDS once.
//
// Arithmetic inside a tile is lazy (Harvey), p < 2^30: a register holds a value < m*p with the
// bound m tracked at compile time (see dft_regs);
//   add:  a + b                                   bound m_a + m_b
//   sub:  a - b + m_b*p                           bound m_a + m_b; usually feeds a twiddle multiply
//   mul by a table twiddle (w, wq = floor(w 2^32 / p)):  a*w - hi(a*wq)*p  -> [0, 2p)  (Shoup)
//   fold: min(x, x - 2p) or min(x, x - 4p), only where the next sum would pass 2^32
// and results are made canonical once, at the store.  The reference's arithmetic is exact
// (`% p` on u128, src/ff.rs:138-160), so canonical residues are bit-identical whatever the
// intermediate representation.
#pragma once
#include "field.h"

// The CPU emulator (tests) checks every claimed bound at run time; device code carries none.
#if defined(SMI_EMU_CHECKS) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#define SMI_BOUND_CHECK(x, m, p) assert((m) <= 8 && (uint64_t)(x) < (uint64_t)((m) ? (m) : 1) * (p) && ((m) || (x) == 0))
#else
#define SMI_BOUND_CHECK(x, m, p) ((void)0)
#endif

#define SMI_TILE_LOG 12          // smallest tile (and the size limit of the single-workgroup kernel)
#define SMI_TILE (1u << SMI_TILE_LOG)
#define SMI_NTT_THREADS 256      // threads of the small kernel; pass kernels use tile/16 threads
#define SMI_TW_LOG 12            // table of w_4096^j as (w, Shoup quotient) pairs: lines of the pass kernels, the small kernel

// A table twiddle in plain form with its Shoup quotient q = floor(w * 2^32 / p).
struct alignas(8) Tw2 {
    uint32_t w, q;
};

struct NttTables {        // per (prime, direction)
    const Tw2 *tw10;      // w_2048^j, j < 2048 (plain + Shoup quotient)
    const uint32_t *lo;   // W^e,          e < 2^h      (Montgomery form; W = primitive 2^K-th root)
    const uint32_t *hi;   // W^(e * 2^h),  e < 2^(K-h)  (Montgomery form)
    uint32_t K, h;
};
struct ScaleTables {      // c * q^i = lo[i & (2^h-1)] * hi[i >> h]   (Montgomery form)
    const uint32_t *lo, *hi;
    uint32_t h;
};
enum { NTT_FIRST = 1, NTT_PRE_SCALE = 2, NTT_POST_SCALE = 4, NTT_TW_SKIP = 8, NTT_TW_IN = 16, NTT_LAST_DIRECT = 32 };

struct PassArgs {
    const uint32_t *in;
    uint32_t *out;
    uint64_t in_stride, out_stride;  // elements between batch columns
    Fp F;
    NttTables T;
    ScaleTables S;
    uint32_t L;        // log2 n
    uint32_t Sp;       // log2 A: digits consumed before this pass
    uint32_t n_in;     // valid inputs of the first pass (zero padded to n)
    uint32_t flags;
    uint32_t d0_log;   // log2 R_0                       (last pass only)
    uint32_t n_mid;    // digits strictly between first and last pass
    uint32_t mid_log[2];
    uint32_t n_tiles;  // grid.x
    uint32_t batch;    // grid.y
    uint32_t pre_ratio_m;   // q^(T*B): step of the input coset scale between a thread's loads (Montgomery)
    uint32_t post_ratio_m;  // q^((R/r_last) << Sp): step of the output scale between a thread's stores (Montgomery)
    uint32_t post_bi_ratio_m;  // q^(r_last << Sp): step of the output scale between a thread's butterflies
    uint32_t zlog;          // first pass: inputs with top digit >= 16 >> zlog are zero padding (0..4)
    // One transform sharded over 2^shard_log ranks (first pass only, 0 otherwise): this rank holds the
    // columns b in [b_off, b_off + B >> shard_log) of the pass's [R][B] view as a strip [R][B >> shard_log];
    // addresses use the strip's row length, twiddles / scales / padding the global index.
    uint32_t shard_log, b_off;
    // NTT_TW_SKIP: this (first) pass stores its outputs lazily reduced (below CAP * p) and leaves its inter-pass twiddle
    // w_m^(k b) to the next pass; NTT_TW_IN: this (middle) pass applies it to its loads -- the first pass
    // of an extension is bound by arithmetic, the middle one by memory.  prev_logr = log2 of the
    // previous pass's digit (k = sub-problem index mod 2^prev_logr).
    uint32_t prev_logr;
};

// digit structure of the in-tile transform; s0 = 4 everywhere, so the 16 values a thread loads in
// a strided pass (rows j0 + i*R/16) are exactly the inputs of its one radix-16 butterfly.
// ntt_pass_cols_kernel (NttPass::share_cols): columns of a tile per workgroup, and the fewest workgroups a launch may
// have for it; the driver's choice to defer the first pass's twiddles (ntt_driver.h) uses the same numbers
enum { SMI_COLS_PER_WG = 4, SMI_COLS_MIN_WGS = 1024 };

template <int LOGR> struct Steps;
template <> struct Steps<6>  { enum { n = 2, s0 = 4, s1 = 2, s2 = 0 }; };
template <> struct Steps<7>  { enum { n = 2, s0 = 4, s1 = 3, s2 = 0 }; };
template <> struct Steps<8>  { enum { n = 2, s0 = 4, s1 = 4, s2 = 0 }; };
template <> struct Steps<9>  { enum { n = 3, s0 = 4, s1 = 3, s2 = 2 }; };
template <> struct Steps<10> { enum { n = 3, s0 = 4, s1 = 3, s2 = 3 }; };
template <> struct Steps<11> { enum { n = 3, s0 = 4, s1 = 4, s2 = 3 }; };

template <int S> SMI_HD constexpr uint32_t brev(uint32_t x) {
    uint32_t r = 0;
    for (int i = 0; i < S; i++) r |= ((x >> i) & 1u) << (S - 1 - i);
    return r;
}

// Blocks are dealt round-robin over the 8 XCDs; tiles that share 128-byte lines are
// adjacent in tile order, so give blocks b, b+8, b+16.. (same XCD, launched together)
// consecutive tiles.  Placement only affects speed (L2 reuse), never results.
SMI_HD uint32_t xcd_tile(uint32_t b, uint32_t n_tiles) {
    if (n_tiles % 8u) return b;
    return (b & 7u) * (n_tiles >> 3) + (b >> 3);
}

SMI_HD uint32_t two_level(const uint32_t *lo, const uint32_t *hi, uint32_t h, uint32_t e, const Fp &F) {
    return mont_mul(lo[e & ((1u << h) - 1u)], hi[e >> h], F);
}

// Global accesses as wave-uniform base + 32-bit byte offset (all tile-relative offsets stay below
// 2^30 elements): lets the compiler use the SGPR-base addressing form instead of a 64-bit
// vector add per access.
#ifndef SMI_NTT_NT
#define SMI_NTT_NT 0   // 1: tuning builds (non-temporal data loads and stores)
#endif
SMI_HD uint32_t ld32(const uint32_t *base, uint32_t idx) {
    const uint32_t *p = (const uint32_t *)((const char *)base + (size_t)(uint32_t)(idx << 2));
#if defined(__HIP_DEVICE_COMPILE__) && SMI_NTT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
SMI_HD void st32(uint32_t *base, uint32_t idx, uint32_t v) {
    uint32_t *p = (uint32_t *)((char *)base + (size_t)(uint32_t)(idx << 2));
#if defined(SMI_NTT_DBG_NOSTORE)   // tuning builds: everything but the store (results are never 2^32-1)
    if (v != 0xFFFFFFFFu) return;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && SMI_NTT_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
SMI_HD Tw2 ld_tw(const Tw2 *base, uint32_t idx) { return *(const Tw2 *)((const char *)base + (size_t)(uint32_t)(idx << 3)); }

// ---- lazy arithmetic on [0, 2p), p < 2^30
SMI_HD uint32_t shoup_mul(uint32_t a, const Tw2 &c, uint32_t p) {               // any a < 2^32 -> [0,2p)
    return a * c.w - umulhi32(a, c.q) * p;
}
SMI_HD uint32_t lz_canon(uint32_t a, uint32_t p) { return umin32(a, a - p); }  // [0,2p) -> [0,p)

// ---- static range bookkeeping.  Every register carries a compile-time bound m: value < m*p.
// Canonical loads have m = 1, Shoup products m = 2, a sum m_u + m_v, a twiddle-free difference
// u - v + m_v*p likewise m_u + m_v.  A value is only folded back when the next butterfly would
// leave [0, 2^32): CAP = 4 for any p < 2^30, CAP = 8 for p < 2^29 (kernels are instantiated for
// both and the launcher picks by the modulus).  After full unrolling the m's are constants, so
// the tests below cost nothing at run time; the CPU emulator evaluates them as ordinary ints.
SMI_HD void lz_fold_m(uint32_t &x, int &m, uint32_t p) {
    if (m > 4) { x = umin32(x, x - 4u * p); m = 4; }          // [0,8p) -> [0,4p)
    else if (m > 2) { x = umin32(x, x - 2u * p); m = 2; }     // [0,4p) -> [0,2p)
}
SMI_HD void lz_fold_to2(uint32_t &x, int &m, uint32_t p) {
    lz_fold_m(x, m, p);
    lz_fold_m(x, m, p);
}
SMI_HD uint32_t lz_canon_m(uint32_t x, int m, uint32_t p) {   // any tracked value -> [0,p)
    lz_fold_to2(x, m, p);
    return m > 1 ? umin32(x, x - p) : x;
}

// S-stage radix-2 DIF on 2^S registers, lazy; x[brev(k)] = X_k on return, m[] updated.  cw = w_R^j
// table in LDS, croot_shift = LOGR - S so that w_r^j = cw[j << croot_shift].
// Z: in the first Z stages the upper input of every butterfly is known to be zero (zero-padded
// transform), so the butterfly degenerates to copy + twiddle.
template <int S, int CAP, int Z = 0>
SMI_HD void dft_regs(uint32_t (&x)[1 << S], int (&m)[1 << S], const Tw2 *cw, int croot_shift, const Fp &F) {
    const uint32_t p = F.p;
#pragma unroll
    for (int s = 0; s < S; s++) {
        const int half = (1 << S) >> (s + 1);
#pragma unroll
        for (int i = 0; i < (1 << S); i++) {
            if (i & half) continue;
            const int j = i | half;
            const int e = (i & (half - 1)) << s;    // w_{2half}^pos = w_r^(pos << s)
            if (s < Z) {
                x[j] = e ? shoup_mul(x[i], cw[e << croot_shift], p) : x[i];
                m[j] = e ? 2 : m[i];
                continue;
            }
#pragma unroll
            for (int t = 0; t < 3; t++)             // at most two folds are ever needed
                if (m[i] + m[j] > CAP) {
                    if (m[i] >= m[j]) lz_fold_m(x[i], m[i], p);
                    else lz_fold_m(x[j], m[j], p);
                }
            const uint32_t u = x[i], v = x[j];
            const int ms = m[i] + m[j];
            SMI_BOUND_CHECK(u, m[i], p);
            SMI_BOUND_CHECK(v, m[j], p);
            x[i] = u + v;                                   // < ms*p <= CAP*p < 2^32
            const uint32_t d = u - v + (uint32_t)m[j] * p;  // (0, ms*p)
            m[i] = ms;
            if (e) { x[j] = shoup_mul(d, cw[e << croot_shift], p); m[j] = 2; }
            else { x[j] = d; m[j] = ms; }
        }
    }
}

// A tile is R = 2^LOGR points x W = 2^LOGW lines, 16 points per thread (NT = R*W/16 threads).
//
// Tile program (sync = workgroup barrier; the emulator runs each phase for every thread first):
//   strided pass: load_tw, load_regs | sync | step0_regs | sync | [step_mid | sync] | last_step_store
//   last pass:    load_tw, load_rows, rows_to_lds | sync | step0_lds | sync | [step_mid | sync] | last_step_store
enum { PASS_FIRST = 0, PASS_MID = 1, PASS_LAST = 2 };   // kernel kinds: first strided pass (zero padding, coset
                                                         // scale), later strided passes, last pass (transposing)
template <int LOGR, int LOGW, int KIND, int CAP> struct NttPass {
    static constexpr bool FIRST = KIND == PASS_FIRST, MID = KIND == PASS_MID, LAST = KIND == PASS_LAST;
    enum { TILE_LOG = LOGR + LOGW, TILE = 1 << TILE_LOG, NT = TILE / 16, R = 1 << LOGR, W = 1 << LOGW, WP = W + 1, V = 16 };
    typedef Steps<LOGR> St;
    enum { SL = St::n == 2 ? St::s1 : St::s2, RL = 1 << SL };   // radix of the last step

    struct TileId {  // wave-uniform description of the tile this workgroup owns
        uint64_t in_base, out_base;  // element offsets of line 0 / output run 0 (within the column)
        uint32_t b0;                 // first inner column (strided passes) / first k_0 (last pass)
    };

    static SMI_HD TileId tile_id(const PassArgs &a, uint32_t block) {
        TileId t;
        const uint32_t tix = xcd_tile(block, a.n_tiles);
        if (!LAST) {
            const uint32_t ablog = a.L - a.Sp - LOGR - a.shard_log;   // log2 of the row length in memory (B, or the strip's)
            const uint32_t tpa = 1u << (ablog - LOGW);                // tiles per sub-problem
            const uint32_t sub = tix / tpa;
            t.b0 = (tix % tpa) << LOGW;
            t.in_base = ((uint64_t)sub << (a.L - a.Sp - a.shard_log)) + t.b0;
            t.out_base = t.in_base;
        } else {
            // lines: k_0 = k0_0 + l (l < W) with the remaining digits a_rest fixed
            const uint32_t g0 = 1u << (a.d0_log - LOGW);   // groups of W adjacent k_0
            const uint32_t a_rest = tix / g0;
            const uint32_t k0_0 = (tix % g0) << LOGW;
            const uint32_t arest_log = a.Sp - a.d0_log;    // log2 (A / R_0)
            // rev'(a_rest): middle digits, most significant first in a_rest, least first in K
            uint32_t rev = 0, shift_in = arest_log, shift_out = 0;
            for (uint32_t d = 0; d < a.n_mid; d++) {
                shift_in -= a.mid_log[d];
                rev |= ((a_rest >> shift_in) & ((1u << a.mid_log[d]) - 1u)) << shift_out;
                shift_out += a.mid_log[d];
            }
            t.b0 = k0_0;
            // line l sits at ((k0_0 + l) * A/R_0 + a_rest) * R: base for l = 0, step (A/R_0)*R
            t.in_base = (((uint64_t)k0_0 << arest_log) + a_rest) << LOGR;
            t.out_base = ((uint64_t)rev << a.d0_log) + k0_0;
        }
        return t;
    }

    static SMI_HD void load_tw(const PassArgs &a, Tw2 *tw, uint32_t tid) {
#pragma unroll
        for (int i = 0; i < (R + NT - 1) / NT; i++) {
            uint32_t j = tid + i * NT;
            if (j < (uint32_t)R) tw[j] = a.T.tw10[j << (SMI_TW_LOG - LOGR)];
        }
    }

    // ---- strided passes: the 16 global loads of a thread are rows j0 + i*(R/16) of column w --
    // exactly the inputs of its radix-16 butterfly (pos = j0), so step 0 runs on them directly.
    // Z: rows i >= 16 >> Z are zero padding and are neither loaded nor scaled.
    // TWIN = false: a caller that holds the deferred twiddles itself (in_mul) applies them after the loads.
    template <int Z, bool TWIN = true> static SMI_HD void load_regs(const PassArgs &a, const TileId &t, uint32_t batch, uint32_t (&v)[V], uint32_t tid) {
        const uint32_t blog = a.L - a.Sp - LOGR, ablog = blog - a.shard_log;
        const uint32_t w = tid & (W - 1), j0 = tid >> LOGW;
        const uint32_t o0 = (j0 << ablog) + w;              // address within the tile's rows
        if constexpr (FIRST) {
            // zero padding: in_base == b0 in the first pass, so b_off + b0 + (j << blog) + w is the natural index.
            // Branch-free (clamped address + select) so the loads issue back to back.
            const uint32_t *col = a.in + (uint64_t)batch * a.in_stride;
            const uint32_t n0 = a.b_off + t.b0 + (j0 << blog) + w;
#pragma unroll
            for (int i = 0; i < V; i++) {
                if (i >= (V >> Z)) { v[i] = 0u; continue; }
                const uint32_t g = n0 + ((uint32_t)(i * (NT >> LOGW)) << blog);
                const uint32_t ga = t.b0 + o0 + ((uint32_t)(i * (NT >> LOGW)) << ablog);
                const uint32_t x = ld32(col, g < a.n_in ? ga : 0u);
                v[i] = g < a.n_in ? x : 0u;
            }
            if (a.flags & NTT_PRE_SCALE) {
                // coset scale q^g: g advances by the constant T<<blog between a thread's loads
                uint32_t sc = two_level(a.S.lo, a.S.hi, a.S.h, n0, a.F);
                const uint32_t rq = a.pre_ratio_m * a.F.pinv;
#pragma unroll
                for (int i = 0; i < (V >> Z); i++) {
                    v[i] = mont_mul(v[i], sc, a.F);
                    sc = mont_mul_c(sc, a.pre_ratio_m, rq, a.F);
                }
            }
        } else {
            const uint32_t *in = a.in + (uint64_t)batch * a.in_stride + t.in_base;
#pragma unroll
            for (int i = 0; i < V; i++) v[i] = ld32(in, o0 + ((uint32_t)(i * (NT >> LOGW)) << ablog));
            if (TWIN && (a.flags & NTT_TW_IN)) {
                // the previous pass's w_m'^(k b'), m' = 2^(L - Sp + prev_logr): k = its output digit = the low
                // prev_logr bits of this tile's sub-problem index, b' = (row << blog) + column; along a
                // thread's rows a geometric sequence whose ratio is the same for the whole tile
                const uint32_t sub = (uint32_t)(t.in_base >> (a.L - a.Sp)), k = sub & ((1u << a.prev_logr) - 1u);
                const uint32_t sh = a.T.K - (a.L - a.Sp + a.prev_logr);
                uint32_t cur = two_level(a.T.lo, a.T.hi, a.T.h, (k * ((j0 << blog) + t.b0 + w)) << sh, a.F);
                const uint32_t ratio = two_level(a.T.lo, a.T.hi, a.T.h, (k * ((uint32_t)(NT >> LOGW) << blog)) << sh, a.F);
                const uint32_t rq = ratio * a.F.pinv;
#pragma unroll
                for (int i = 0; i < V; i++) {
                    v[i] = mont_mul(v[i], cur, a.F);
                    if (i + 1 < V) cur = mont_mul_c(cur, ratio, rq, a.F);
                }
            }
        }
    }

    // radix-16 step 0 on registers (strided passes); writes the tile to LDS
    template <int Z> static SMI_HD void step0_regs(const PassArgs &a, uint32_t (&x)[V], uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        enum { SUB = LOGR - 4 };
        const uint32_t w = tid & (W - 1), pos = tid >> LOGW;
        int m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = i < (V >> Z) ? 1 : 0;   // canonical loads / scaled products; padding
        dft_regs<4, CAP, Z>(x, m, tw, LOGR - 4, a.F);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            uint32_t v = x[brev<4>(kk)];
            int mv = m[brev<4>(kk)];
            if (kk) v = shoup_mul(v, tw[(pos * kk) & (R - 1)], a.F.p);
            else lz_fold_to2(v, mv, a.F.p);
            tile[(pos + ((uint32_t)kk << SUB)) * WP + w] = v;       // LDS holds [0,2p)
        }
    }

    // ---- last pass: rows are contiguous in HBM; load coalesced along the row, transpose via LDS.
    // Two phases (global -> registers, registers -> LDS): all 16 loads are in flight together.
    static SMI_HD void load_rows(const PassArgs &a, const TileId &t, uint32_t batch, uint32_t (&v)[V], uint32_t tid) {
        const uint32_t *in = a.in + (uint64_t)batch * a.in_stride + t.in_base;
        const uint32_t arest_log = a.Sp - a.d0_log;
#pragma unroll
        for (int i = 0; i < V; i++) {
            const uint32_t idx = tid + i * NT;
            const uint32_t j = idx & (R - 1), l = idx >> LOGR;
            v[i] = ld32(in, (l << (arest_log + LOGR)) + j);
        }
    }
    static SMI_HD void rows_to_lds(const uint32_t (&v)[V], uint32_t *tile, uint32_t tid) {
#pragma unroll
        for (int i = 0; i < V; i++) {
            const uint32_t idx = tid + i * NT;
            tile[(idx & (R - 1)) * WP + (idx >> LOGR)] = v[i];
        }
    }

    // ---- last pass, NTT_LAST_DIRECT: step 0 on the registers the loads landed in, like the strided passes -- no
    // transposition through LDS before the first butterflies (one LDS round trip and one barrier fewer).  A thread owns
    // (pos, line) with pos the fastest index over the lanes, so its 16 loads are elements pos + q*R/16 of its line
    // (R/16 consecutive elements per line and load instruction: 64-byte pieces at R = 256) and exactly the inputs of
    // its radix-16 butterfly.  Lines are dealt to the lanes rotated by log2(R/16) bits so that the LDS writes of a
    // half-wave (row stride WP = W + 1, bank = (pos + line) mod 32) fall into distinct banks.
    static SMI_HD void direct_map(uint32_t tid, uint32_t &pos, uint32_t &line) {
        enum { SUB = LOGR - 4, ROT = SUB >= 5 ? 0 : SUB % LOGW };
        pos = tid & ((1u << SUB) - 1u);
        const uint32_t lsel = tid >> SUB;
        if constexpr (ROT != 0) line = ((lsel << ROT) | (lsel >> (LOGW - ROT))) & (W - 1);
        else line = lsel;
    }
    static SMI_HD void load_rows_direct(const PassArgs &a, const TileId &t, uint32_t batch, uint32_t (&v)[V], uint32_t tid) {
        enum { SUB = LOGR - 4 };
        const uint32_t *in = a.in + (uint64_t)batch * a.in_stride + t.in_base;
        const uint32_t arest_log = a.Sp - a.d0_log;
        uint32_t pos, line;
        direct_map(tid, pos, line);
        const uint32_t o0 = (line << (arest_log + LOGR)) + pos;
#pragma unroll
        for (int i = 0; i < V; i++) v[i] = ld32(in, o0 + ((uint32_t)i << SUB));
    }
    static SMI_HD void step0_rows(const PassArgs &a, uint32_t (&x)[V], uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        enum { SUB = LOGR - 4 };
        uint32_t pos, line;
        direct_map(tid, pos, line);
        int m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = 1;                      // canonical loads
        dft_regs<4, CAP>(x, m, tw, LOGR - 4, a.F);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            uint32_t v = x[brev<4>(kk)];
            int mv = m[brev<4>(kk)];
            if (kk) v = shoup_mul(v, tw[(pos * kk) & (R - 1)], a.F.p);
            else lz_fold_to2(v, mv, a.F.p);
            tile[(pos + ((uint32_t)kk << SUB)) * WP + line] = v;    // what step0_lds leaves: LDS holds [0,2p)
        }
    }

    // MIN: bound of the values read (1 = canonical, straight from a global load; 2 = earlier step).
    template <int S, int MLOG, int MIN> static SMI_HD void step(const PassArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        enum { r = 1 << S, SUB = MLOG - S, NB = (TILE / r) / NT };
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const uint32_t u = tid + bi * NT;
            const uint32_t w = u & (W - 1), ub = u >> LOGW;
            const uint32_t blk = ub >> SUB, pos = ub & ((1u << SUB) - 1u);
            const uint32_t base = (blk << MLOG) + pos;
            uint32_t x[r];
#pragma unroll
            for (int q = 0; q < r; q++) x[q] = tile[(base + ((uint32_t)q << SUB)) * WP + w];
            int m[r];
#pragma unroll
            for (int q = 0; q < r; q++) m[q] = MIN;
            dft_regs<S, CAP>(x, m, tw, LOGR - S, a.F);
#pragma unroll
            for (int kk = 0; kk < r; kk++) {
                uint32_t v = x[brev<S>(kk)];
                int mv = m[brev<S>(kk)];
                if (kk) v = shoup_mul(v, tw[((pos * kk) << (LOGR - MLOG)) & (R - 1)], a.F.p);
                else lz_fold_to2(v, mv, a.F.p);
                tile[(base + ((uint32_t)kk << SUB)) * WP + w] = v;
            }
        }
    }
    static SMI_HD void step0_lds(const PassArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) { step<4, LOGR, 1>(a, tile, tw, tid); }
    // the middle step of a three-step tile (no-op for two-step tiles)
    static SMI_HD void step_mid(const PassArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        if constexpr (St::n == 3) step<St::s1, LOGR - St::s0, 2>(a, tile, tw, tid);
    }

    // position of a last-step block (all digits but the last) -> its weight in the frequency index
    static SMI_HD uint32_t blk_to_k(uint32_t blk) {
        if (St::n == 2) return blk;                                  // k0
        const uint32_t k0 = blk >> St::s1, k1 = blk & ((1u << St::s1) - 1u);
        return k0 | (k1 << St::s0);
    }

    // Last in-tile step (radix RL, no step twiddles) fused with the store: outputs leave from
    // registers.  A butterfly's outputs are frequencies k = kbase + kk*(R/RL), kk < RL: an
    // arithmetic progression, so inter-pass twiddles / output scales are running products.
    static SMI_HD void last_step_store(const PassArgs &a, const TileId &t, uint32_t batch, const uint32_t *tile, const Tw2 *tw,
                                       uint32_t tid) {
        enum { NB = (TILE / RL) / NT, KSTEP_LOG = LOGR - SL };
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride + t.out_base;
        const uint32_t mlog = a.L - a.Sp, p = a.F.p;
        // Butterfly bi of a thread has kbase = kbase_0 + bi * RL (its block index advances by R/16,
        // i.e. k_0 by RL), so the per-butterfly bases are running products as well: three table
        // lookups per thread whatever NB is.
        uint32_t base_run = 0, gs = 0, gq = 0, gbi = 0, gbq = 0;   // first pass: g^kbase, g^(R/RL), g^RL
        uint32_t sc_run = 0, rq = 0, rbq = 0;                        // last pass: scale(kn_base) and ratios
        if (!LAST && !(a.flags & NTT_TW_SKIP)) {
            const uint32_t b = a.b_off + t.b0 + (tid & (W - 1)), sh = a.T.K - mlog;
            base_run = two_level(a.T.lo, a.T.hi, a.T.h, (b * blk_to_k(tid >> LOGW)) << sh, a.F);
            gs = two_level(a.T.lo, a.T.hi, a.T.h, (b << KSTEP_LOG) << sh, a.F);
            gq = gs * a.F.pinv;
            if (NB > 1) {
                gbi = two_level(a.T.lo, a.T.hi, a.T.h, (b << SL) << sh, a.F);
                gbq = gbi * a.F.pinv;
            }
        }
        if (LAST && (a.flags & NTT_POST_SCALE)) {
            sc_run = two_level(a.S.lo, a.S.hi, a.S.h, (uint32_t)t.out_base + (blk_to_k(tid >> LOGW) << a.Sp) + (tid & (W - 1)), a.F);
            rq = a.post_ratio_m * a.F.pinv;
            rbq = a.post_bi_ratio_m * a.F.pinv;
        }
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const uint32_t u = tid + bi * NT;
            const uint32_t w = u & (W - 1), blk = u >> LOGW;
            uint32_t x[RL];
#pragma unroll
            for (int q = 0; q < RL; q++) x[q] = tile[(blk * RL + q) * WP + w];
            int m[RL];
#pragma unroll
            for (int q = 0; q < RL; q++) m[q] = 2;
            dft_regs<SL, CAP>(x, m, tw, LOGR - SL, a.F);   // outputs < CAP*p: fine for any multiply below
            const uint32_t kbase = blk_to_k(blk);
            if constexpr (!LAST) {
                const uint32_t blog = mlog - LOGR - a.shard_log;   // row length in memory
                const uint32_t o0 = (kbase << blog) + w;
                {
                    // inter-pass twiddles w_m^(k*b) = g^k, g = w_m^b: running products over kk (and over
                    // the thread's butterflies, see above).  A per-pass table of (w, Shoup quotient)
                    // pairs read with the data's coalescing saves 60 VALU ops per thread but triples
                    // the L2 -> L1 traffic of the pass; measured slower (2^25 x 4: 266 vs 247 us).
                    if (a.flags & NTT_TW_SKIP) {
                        // The next pass multiplies as it loads, by a Montgomery product that takes any operand below
                        // 2^32 beside a canonical one (a * b < p * 2^32 <=> CAP * p < 2^32): the values go to the
                        // pass-private intermediate as they are, below CAP * p, without the folds back to [0, p)
                        // (up to six instructions per output at CAP = 8).
#pragma unroll
                        for (int kk = 0; kk < RL; kk++) {
                            SMI_BOUND_CHECK(x[brev<SL>(kk)], m[brev<SL>(kk)], p);
                            st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + blog)), x[brev<SL>(kk)]);
                        }
                        continue;
                    }
                    uint32_t cur = base_run;
                    if (bi + 1 < NB) base_run = mont_mul_c(base_run, gbi, gbq, a.F);
#pragma unroll
                    for (int kk = 0; kk < RL; kk++) {
                        st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + blog)), mont_mul(x[brev<SL>(kk)], cur, a.F));
                        if (kk + 1 < RL) cur = mont_mul_c(cur, gs, gq, a.F);
                    }
                }
            } else {
                const uint32_t o0 = (kbase << a.Sp) + w;
                if (a.flags & NTT_POST_SCALE) {
                    uint32_t sc = sc_run;
                    if (bi + 1 < NB) sc_run = mont_mul_c(sc_run, a.post_bi_ratio_m, rbq, a.F);
#pragma unroll
                    for (int kk = 0; kk < RL; kk++) {
                        st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + a.Sp)), mont_mul(x[brev<SL>(kk)], sc, a.F));
                        if (kk + 1 < RL) sc = mont_mul_c(sc, a.post_ratio_m, rq, a.F);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < RL; kk++) st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + a.Sp)), lz_canon_m(x[brev<SL>(kk)], m[brev<SL>(kk)], p));
                }
            }
        }
    }

    // ---- all columns of a tile in one workgroup (ntt_pass_kernel's SHARE).  The multipliers of a thread's 16 outputs
    // -- the inter-pass twiddles of a strided pass, the output scale of a last pass -- depend on the tile and the
    // thread, not on the column: derived once, in the order last_step_store applies them, with their companions
    // w * p^-1 so that every application is a three-multiply mont_mul_c.  The products are the canonical residues
    // last_step_store writes.
    static SMI_HD bool has_out_mul(const PassArgs &a) { return LAST ? (a.flags & NTT_POST_SCALE) != 0 : !(a.flags & NTT_TW_SKIP); }
    // The launchers' rule.  Not the first pass: measured on MI355X (gpurun_out/exp_share_cols.log) its 2^14-point
    // tiles lose more to the registers the multipliers occupy (one workgroup per CU instead of two) or to the
    // serialised columns than the saved products give back (2^25 x 4: 205-210 us against 197).
    // At most COLS_PER_WG columns per workgroup (grid.y = the column groups), and only while the launch still has
    // enough workgroups to fill the chip -- many short columns keep one workgroup per (tile, column).
    enum { COLS_PER_WG = SMI_COLS_PER_WG, COLS_MIN_WGS = SMI_COLS_MIN_WGS };
    static SMI_HD uint32_t col_groups(const PassArgs &a) { return (a.batch + COLS_PER_WG - 1) / COLS_PER_WG; }
    static SMI_HD bool share_cols(const PassArgs &a, uint32_t min_wgs = COLS_MIN_WGS) {
        return !FIRST && a.batch > 1 && has_out_mul(a) && (uint64_t)a.n_tiles * col_groups(a) >= min_wgs;
    }
    // MQ = false: only the multipliers are kept (16 fewer VGPRs, one multiply more per output).
    template <bool MQ = true>
    static SMI_HD void out_mul(const PassArgs &a, const TileId &t, uint32_t tid, uint32_t (&mw)[V], uint32_t (&mq)[MQ ? V : 1]) {
        enum { NB = (TILE / RL) / NT, KSTEP_LOG = LOGR - SL };
        const uint32_t w = tid & (W - 1), kb0 = blk_to_k(tid >> LOGW);
        uint32_t run, step, step_bi = 0;
        if (!LAST) {
            const uint32_t b = a.b_off + t.b0 + w, sh = a.T.K - (a.L - a.Sp);
            run = two_level(a.T.lo, a.T.hi, a.T.h, (b * kb0) << sh, a.F);
            step = two_level(a.T.lo, a.T.hi, a.T.h, (b << KSTEP_LOG) << sh, a.F);
            if (NB > 1) step_bi = two_level(a.T.lo, a.T.hi, a.T.h, (b << SL) << sh, a.F);
        } else {
            run = two_level(a.S.lo, a.S.hi, a.S.h, (uint32_t)t.out_base + (kb0 << a.Sp) + w, a.F);
            step = a.post_ratio_m;
            step_bi = a.post_bi_ratio_m;
        }
        const uint32_t sq = step * a.F.pinv, bq = step_bi * a.F.pinv;
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            uint32_t cur = run;
            if (bi + 1 < NB) run = mont_mul_c(run, step_bi, bq, a.F);
#pragma unroll
            for (int kk = 0; kk < RL; kk++) {
                mw[bi * RL + kk] = cur;
                if constexpr (MQ) mq[bi * RL + kk] = cur * a.F.pinv;
                if (kk + 1 < RL) cur = mont_mul_c(cur, step, sq, a.F);
            }
        }
    }
    // The deferred twiddles of load_regs (NTT_TW_IN) as the thread's 16 input multipliers: like the output multipliers
    // they depend on the tile and the thread only.  Same sequence as load_regs derives on the fly.
    static SMI_HD void in_mul(const PassArgs &a, const TileId &t, uint32_t tid, uint32_t (&iw)[V]) {
        const uint32_t blog = a.L - a.Sp - LOGR;
        const uint32_t w = tid & (W - 1), j0 = tid >> LOGW;
        const uint32_t sub = (uint32_t)(t.in_base >> (a.L - a.Sp)), k = sub & ((1u << a.prev_logr) - 1u);
        const uint32_t sh = a.T.K - (a.L - a.Sp + a.prev_logr);
        uint32_t cur = two_level(a.T.lo, a.T.hi, a.T.h, (k * ((j0 << blog) + t.b0 + w)) << sh, a.F);
        const uint32_t ratio = two_level(a.T.lo, a.T.hi, a.T.h, (k * ((uint32_t)(NT >> LOGW) << blog)) << sh, a.F);
        const uint32_t rq = ratio * a.F.pinv;
#pragma unroll
        for (int i = 0; i < V; i++) {
            iw[i] = cur;
            if (i + 1 < V) cur = mont_mul_c(cur, ratio, rq, a.F);
        }
    }
    template <bool MQ = true>
    static SMI_HD void last_step_store_mul(const PassArgs &a, const TileId &t, uint32_t batch, const uint32_t *tile, const Tw2 *tw,
                                           uint32_t tid, const uint32_t (&mw)[V], const uint32_t (&mq)[MQ ? V : 1]) {
        enum { NB = (TILE / RL) / NT, KSTEP_LOG = LOGR - SL };
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride + t.out_base;
        const uint32_t olog = LAST ? a.Sp : a.L - a.Sp - LOGR - a.shard_log;   // log2 of the frequency index's stride in memory
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const uint32_t u = tid + bi * NT;
            const uint32_t w = u & (W - 1), blk = u >> LOGW;
            uint32_t x[RL];
#pragma unroll
            for (int q = 0; q < RL; q++) x[q] = tile[(blk * RL + q) * WP + w];
            int m[RL];
#pragma unroll
            for (int q = 0; q < RL; q++) m[q] = 2;
            dft_regs<SL, CAP>(x, m, tw, LOGR - SL, a.F);
            const uint32_t o0 = (blk_to_k(blk) << olog) + w;
#pragma unroll
            for (int kk = 0; kk < RL; kk++)
                if constexpr (MQ) st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + olog)), mont_mul_c(x[brev<SL>(kk)], mw[bi * RL + kk], mq[bi * RL + kk], a.F));
                else st32(out, o0 + ((uint32_t)kk << (KSTEP_LOG + olog)), mont_mul(x[brev<SL>(kk)], mw[bi * RL + kk], a.F));
        }
    }
};

// ---- small transforms (n <= 4096): one workgroup per column, radix-4 DIF stages in LDS (a radix-2
// stage when log n is odd), twiddles w_n^j staged in LDS once per workgroup, bit-reversed read at
// the store.  Latency-sized work: the pass kernels above take over from n = 8192.
struct SmallArgs {
    const uint32_t *in;
    uint32_t *out;
    uint64_t in_stride, out_stride;
    Fp F;
    NttTables T;
    ScaleTables S;
    uint32_t L, n_in, flags;
};
struct NttSmall {
    // LDS index of point i: one word of padding per 64, so that the bit-reversed read of the store
    // (addresses 2^(L-6) apart across a wave) and the short-span stages spread over the banks
    static SMI_HD uint32_t at(uint32_t i) { return i + (i >> 6); }
    // twm[j] = w_n^j with its Shoup quotient, j < n/2 (a stride of the context's w_4096^j table)
    static SMI_HD void load_tw(const SmallArgs &a, Tw2 *twm, uint32_t tid) {
        const uint32_t half_n = (1u << a.L) >> 1;
        for (uint32_t j = tid; j < half_n; j += SMI_NTT_THREADS) twm[j] = a.T.tw10[j << (SMI_TW_LOG - a.L)];
    }
    static SMI_HD void load(const SmallArgs &a, uint32_t batch, uint32_t *buf, uint32_t tid) {
        const uint32_t n = 1u << a.L;
        const uint32_t *in = a.in + (uint64_t)batch * a.in_stride;
        for (uint32_t g = tid; g < n; g += SMI_NTT_THREADS) {
            uint32_t v = g < a.n_in ? in[g] : 0u;
            if ((a.flags & NTT_PRE_SCALE) && v) v = mont_mul(v, two_level(a.S.lo, a.S.hi, a.S.h, g, a.F), a.F);
            buf[at(g)] = v;
        }
    }
    // x*w mod p, canonical (Shoup product in [0,2p), one conditional subtraction)
    static SMI_HD uint32_t mulw(uint32_t x, const Tw2 &w, uint32_t p) { return lz_canon(shoup_mul(x, w, p), p); }
    // stages s and s+1 at once (spans h = n >> (s+1) and h/2): one radix-4 butterfly per 4 points
    static SMI_HD void stage4(const SmallArgs &a, uint32_t s, uint32_t *buf, const Tw2 *twm, uint32_t tid) {
        const uint32_t n = 1u << a.L, qlog = a.L - s - 2, q = 1u << qlog, p = a.F.p;   // q = h/2
        for (uint32_t u = tid; u < n / 4; u += SMI_NTT_THREADS) {
            const uint32_t blk = u >> qlog, pos = u & (q - 1);
            const uint32_t i = (blk << (qlog + 2)) + pos;
            const uint32_t x0 = buf[at(i)], x1 = buf[at(i + q)], x2 = buf[at(i + 2 * q)], x3 = buf[at(i + 3 * q)];
            // stage s: (x0, x2) with w_2h^pos, (x1, x3) with w_2h^(pos + q)
            const uint32_t a0 = fp_add(x0, x2, p), a1 = fp_add(x1, x3, p);
            const uint32_t c0 = mulw(fp_sub(x0, x2, p), twm[pos << s], p);
            const uint32_t c1 = mulw(fp_sub(x1, x3, p), twm[(pos + q) << s], p);
            // stage s+1: (a0, a1) and (c0, c1) with w_h^pos
            const Tw2 w = twm[pos << (s + 1)];
            buf[at(i)] = fp_add(a0, a1, p);
            buf[at(i + q)] = mulw(fp_sub(a0, a1, p), w, p);
            buf[at(i + 2 * q)] = fp_add(c0, c1, p);
            buf[at(i + 3 * q)] = mulw(fp_sub(c0, c1, p), w, p);
        }
    }
    // a single stage s: butterflies of span half = n >> (s+1)
    static SMI_HD void stage(const SmallArgs &a, uint32_t s, uint32_t *buf, const Tw2 *twm, uint32_t tid) {
        const uint32_t n = 1u << a.L, hlog = a.L - s - 1, half = 1u << hlog;
        for (uint32_t u = tid; u < n / 2; u += SMI_NTT_THREADS) {
            const uint32_t blk = u >> hlog, pos = u & (half - 1);
            const uint32_t i = (blk << (hlog + 1)) + pos, j = i + half;
            const uint32_t x = buf[at(i)], y = buf[at(j)];
            buf[at(i)] = fp_add(x, y, a.F.p);
            buf[at(j)] = mulw(fp_sub(x, y, a.F.p), twm[pos << s], a.F.p);
        }
    }
    static SMI_HD void store(const SmallArgs &a, uint32_t batch, const uint32_t *buf, uint32_t tid) {
        const uint32_t n = 1u << a.L;
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride;
        for (uint32_t k = tid; k < n; k += SMI_NTT_THREADS) {
            uint32_t v = buf[at(a.L ? bitrev32(k) >> (32u - a.L) : 0u)];
            if (a.flags & NTT_POST_SCALE) v = mont_mul(v, two_level(a.S.lo, a.S.hi, a.S.h, k, a.F), a.F);
            out[k] = v;
        }
    }
};
