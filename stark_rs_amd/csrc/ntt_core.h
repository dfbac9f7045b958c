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
// One workgroup = 256 threads = one 4096-element tile = W adjacent lines of R points
// (R*W = 4096), staged in LDS as [R][W+1].  Lines are W adjacent b-columns (runs of W*4
// contiguous bytes in HBM) for strided passes, and W rows with adjacent k_0 for the last
// pass (so its transposed writes are W-element runs too).  Inside the tile the R-point
// transform is 2-3 register-resident radix-8/16 steps with an LDS exchange between them;
// twiddles w_R^j are staged in LDS once per workgroup.
#pragma once
#include "field.h"

#define SMI_TILE_LOG 12          // smallest tile (and the size limit of the single-workgroup kernel)
#define SMI_TILE (1u << SMI_TILE_LOG)
#define SMI_NTT_THREADS 256      // threads of the small kernel; pass kernels use tile/16 threads
#define SMI_TW_LOG 10   // in-tile twiddle table: w_1024^j

// A table twiddle with its Montgomery companion q = w * p^-1 mod 2^32 (see mont_mul_c).
struct alignas(8) Tw2 {
    uint32_t w, q;
};

struct NttTables {        // per (prime, direction); all values in Montgomery form
    const Tw2 *tw10;      // w_1024^j, j < 1024
    const uint32_t *lo;   // W^e,          e < 2^h      (W = primitive 2^K-th root for this direction)
    const uint32_t *hi;   // W^(e * 2^h),  e < 2^(K-h)
    uint32_t K, h;
};
struct ScaleTables {      // c * q^i = lo[i & (2^h-1)] * hi[i >> h]   (Montgomery form)
    const uint32_t *lo, *hi;
    uint32_t h;
};
enum { NTT_FIRST = 1, NTT_PRE_SCALE = 2, NTT_POST_SCALE = 4 };

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
    const Tw2 *ptab;        // inter-pass twiddles w_m^(k*b) at [k*B + b] for passes after the first (else null)
    uint32_t pre_ratio_m;   // q^(T*B): step of the input coset scale between a thread's loads (Montgomery)
    uint32_t post_ratio_m;  // q^(2^Sp): step of the output scale between a thread's stores (Montgomery)
};

// digit structure of the in-tile transform.  s0 = 4 everywhere: in the store phase thread t owns
// column w = t % W and the 16 positions loc = (t / W) + i * (256 / W), whose leading digit is i, so
// its frequencies are k = i + 16 * krest -- consecutive, which lets inter-pass twiddles and coset
// scales be running products instead of per-element table gathers.
template <int LOGR> struct Steps;
template <> struct Steps<6>  { enum { n = 2, s0 = 4, s1 = 2, s2 = 0 }; };
template <> struct Steps<7>  { enum { n = 2, s0 = 4, s1 = 3, s2 = 0 }; };
template <> struct Steps<8>  { enum { n = 2, s0 = 4, s1 = 4, s2 = 0 }; };
template <> struct Steps<9>  { enum { n = 3, s0 = 4, s1 = 3, s2 = 2 }; };
template <> struct Steps<10> { enum { n = 3, s0 = 4, s1 = 3, s2 = 3 }; };

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

// S-stage radix-2 DIF on 2^S registers; x[brev(k)] = X_k on return.  cw = w_R^j table in
// LDS, croot_shift = LOGR - S so that w_r^j = cw[j << croot_shift].
template <int S> SMI_HD void dft_regs(uint32_t (&x)[1 << S], const Tw2 *cw, int croot_shift, const Fp &F) {
#pragma unroll
    for (int s = 0; s < S; s++) {
        const int half = (1 << S) >> (s + 1);
#pragma unroll
        for (int i = 0; i < (1 << S); i++) {
            if (i & half) continue;
            const int j = i | half;
            uint32_t u = x[i], v = x[j];
            x[i] = fp_add(u, v, F.p);
            uint32_t d = fp_sub(u, v, F.p);
            const int e = (i & (half - 1)) << s;  // w_{2half}^pos = w_r^(pos << s)
            if (e) {
                const Tw2 c = cw[e << croot_shift];
                d = mont_mul_c(d, c.w, c.q, F);
            }
            x[j] = d;
        }
    }
}

// A tile is R = 2^LOGR points x W = 2^LOGW lines, 16 points per thread (NT = R*W/16 threads).
template <int LOGR, int LOGW, bool LAST> struct NttPass {
    enum { TILE_LOG = LOGR + LOGW, TILE = 1 << TILE_LOG, NT = TILE / 16, R = 1 << LOGR, W = 1 << LOGW, WP = W + 1, V = 16 };
    typedef Steps<LOGR> St;

    struct TileId {  // wave-uniform description of the tile this workgroup owns
        uint64_t in_base, out_base;  // element offsets of line 0 / output run 0 (within the column)
        uint32_t b0;                 // first inner column (strided passes)
    };

    static SMI_HD TileId tile_id(const PassArgs &a, uint32_t block) {
        TileId t;
        const uint32_t tix = xcd_tile(block, a.n_tiles);
        if (!LAST) {
            const uint32_t blog = a.L - a.Sp - LOGR;       // log2 B
            const uint32_t tpa = 1u << (blog - LOGW);      // tiles per sub-problem
            const uint32_t sub = tix / tpa;
            t.b0 = (tix % tpa) << LOGW;
            t.in_base = ((uint64_t)sub << (a.L - a.Sp)) + t.b0;
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

    static SMI_HD void load(const PassArgs &a, const TileId &t, uint32_t batch, uint32_t *tile, uint32_t tid) {
        // wave-uniform base pointer + 32-bit per-lane offsets (tile offsets stay below 2^27 elements)
        const uint32_t *in = a.in + (uint64_t)batch * a.in_stride + t.in_base;
        if (!LAST) {
            const uint32_t blog = a.L - a.Sp - LOGR;
            // thread-constant column w; rows j = j0 + i*T.  The coset scale q^g of input index
            // g = g0 + i*(T<<blog) advances by the constant pre_ratio = q^(T<<blog).
            const uint32_t w = tid & (W - 1), j0 = tid >> LOGW;
            const uint32_t o0 = (j0 << blog) + w;
            uint32_t v[V];
            if (a.flags & NTT_FIRST) {
                // zero padding: in_base == b0 in the first pass, so b0 + o is the natural index.
                // Branch-free (clamped address + select) so the 16 loads issue back to back.
                const uint32_t *col = a.in + (uint64_t)batch * a.in_stride;
#pragma unroll
                for (int i = 0; i < V; i++) {
                    const uint32_t g = t.b0 + o0 + ((uint32_t)(i * (NT >> LOGW)) << blog);
                    const uint32_t x = col[g < a.n_in ? g : 0u];
                    v[i] = g < a.n_in ? x : 0u;
                }
                if (a.flags & NTT_PRE_SCALE) {
                    uint32_t sc = two_level(a.S.lo, a.S.hi, a.S.h, t.b0 + o0, a.F);
                    const uint32_t rq = a.pre_ratio_m * a.F.pinv;
#pragma unroll
                    for (int i = 0; i < V; i++) {
                        v[i] = mont_mul(v[i], sc, a.F);
                        sc = mont_mul_c(sc, a.pre_ratio_m, rq, a.F);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < V; i++) v[i] = in[o0 + ((uint32_t)(i * (NT >> LOGW)) << blog)];
            }
#pragma unroll
            for (int i = 0; i < V; i++) tile[(j0 + i * (NT >> LOGW)) * WP + w] = v[i];
        } else {
            const uint32_t arest_log = a.Sp - a.d0_log;
            uint32_t v[V];
#pragma unroll
            for (int i = 0; i < V; i++) {
                const uint32_t idx = tid + i * NT;
                const uint32_t j = idx & (R - 1), l = idx >> LOGR;
                v[i] = in[(l << (arest_log + LOGR)) + j];
            }
#pragma unroll
            for (int i = 0; i < V; i++) {
                const uint32_t idx = tid + i * NT;
                tile[(idx & (R - 1)) * WP + (idx >> LOGR)] = v[i];
            }
        }
    }

    // One radix-2^S step on sub-blocks of 2^MLOG points (MLOG = log2 M of this step).
    template <int S, int MLOG> static SMI_HD void step(const PassArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
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
            dft_regs<S>(x, tw, LOGR - S, a.F);
#pragma unroll
            for (int kk = 0; kk < r; kk++) {
                uint32_t v = x[brev<S>(kk)];
                if (SUB > 0 && kk) {
                    const Tw2 c = tw[((pos * kk) << (LOGR - MLOG)) & (R - 1)];
                    v = mont_mul_c(v, c.w, c.q, a.F);
                }
                tile[(base + ((uint32_t)kk << SUB)) * WP + w] = v;
            }
        }
    }

    // step I of the in-tile transform (I < St::n)
    template <int I> static SMI_HD void step_i(const PassArgs &a, uint32_t *tile, const Tw2 *tw, uint32_t tid) {
        if constexpr (I == 0) step<St::s0, LOGR>(a, tile, tw, tid);
        else if constexpr (I == 1) step<St::s1, LOGR - St::s0>(a, tile, tw, tid);
        else if constexpr (I == 2 && St::n == 3) step<St::s2, LOGR - St::s0 - St::s1>(a, tile, tw, tid);
    }

    // position in the line after all steps -> natural in-line frequency index
    static SMI_HD uint32_t loc_to_k(uint32_t loc) {
        if (St::n == 2) {
            const uint32_t k0 = loc >> St::s1, k1 = loc & ((1u << St::s1) - 1u);
            return k0 | (k1 << St::s0);
        } else {
            const uint32_t k0 = loc >> (St::s1 + St::s2);
            const uint32_t k1 = (loc >> St::s2) & ((1u << St::s1) - 1u);
            const uint32_t k2 = loc & ((1u << St::s2) - 1u);
            return k0 | (k1 << St::s0) | (k2 << (St::s0 + St::s1));
        }
    }

    // rest digits of a position (everything below the leading radix-16 digit) -> their weight in k
    static SMI_HD uint32_t rest_to_k(uint32_t loc0) {
        if (St::n == 2) return loc0;  // k1
        const uint32_t k1 = loc0 >> St::s2, k2 = loc0 & ((1u << St::s2) - 1u);
        return k1 | (k2 << St::s1);
    }

    static SMI_HD void store(const PassArgs &a, const TileId &t, uint32_t batch, const uint32_t *tile, uint32_t tid) {
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride + t.out_base;
        const uint32_t mlog = a.L - a.Sp;  // log2 m (sub-problem size of this pass)
        // thread-constant column w; positions loc = loc0 + i*T have leading digit i, so the
        // frequencies are k = i + 16*krest, i = 0..15.
        const uint32_t w = tid & (W - 1), loc0 = tid >> LOGW;
        const uint32_t krest = rest_to_k(loc0);
        if (!LAST) {
            const uint32_t blog = mlog - LOGR;
            const uint32_t b = t.b0 + w;
            const uint32_t o0 = ((krest << 4) << blog) + w;
            if (a.ptab) {
                // passes after the first: w_m^(k*b) from the (L2-resident) table of this pass,
                // read with the same coalescing as the data
                const Tw2 *tab = a.ptab + t.b0;
#pragma unroll
                for (int i = 0; i < V; i++) {
                    const uint32_t loc = loc0 + i * (NT >> LOGW);
                    const uint32_t o = o0 + ((uint32_t)i << blog);
                    const Tw2 c = tab[o];
                    out[o] = mont_mul_c(tile[loc * WP + w], c.w, c.q, a.F);
                }
            } else {
                // first pass (m = n: a table would double the traffic): w_m^(k*b) = g^k with
                // g = w_m^b; two lookups per thread, then a running product over consecutive k.
                const uint32_t sh = a.T.K - mlog;
                const uint32_t g = two_level(a.T.lo, a.T.hi, a.T.h, b << sh, a.F);
                const uint32_t gq = g * a.F.pinv;
                uint32_t cur = two_level(a.T.lo, a.T.hi, a.T.h, (b * (krest << 4)) << sh, a.F);
#pragma unroll
                for (int i = 0; i < V; i++) {
                    const uint32_t loc = loc0 + i * (NT >> LOGW);
                    const uint32_t v = mont_mul(tile[loc * WP + w], cur, a.F);
                    cur = mont_mul_c(cur, g, gq, a.F);
                    out[o0 + ((uint32_t)i << blog)] = v;
                }
            }
        } else {
            const uint32_t o0 = ((krest << 4) << a.Sp) + w;
            uint32_t sc = 0;
            if (a.flags & NTT_POST_SCALE) sc = two_level(a.S.lo, a.S.hi, a.S.h, (uint32_t)t.out_base + o0, a.F);
            const uint32_t rq = a.post_ratio_m * a.F.pinv;
#pragma unroll
            for (int i = 0; i < V; i++) {
                const uint32_t loc = loc0 + i * (NT >> LOGW);
                uint32_t v = tile[loc * WP + w];
                if (a.flags & NTT_POST_SCALE) {
                    v = mont_mul(v, sc, a.F);
                    sc = mont_mul_c(sc, a.post_ratio_m, rq, a.F);
                }
                out[o0 + ((uint32_t)i << a.Sp)] = v;
            }
        }
    }
};

// ---- small transforms (n <= 4096): one workgroup per column, radix-2 DIF in LDS --------
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
    static SMI_HD void load(const SmallArgs &a, uint32_t batch, uint32_t *buf, uint32_t tid) {
        const uint32_t n = 1u << a.L;
        const uint32_t *in = a.in + (uint64_t)batch * a.in_stride;
        for (uint32_t g = tid; g < n; g += SMI_NTT_THREADS) {
            uint32_t v = g < a.n_in ? in[g] : 0u;
            if ((a.flags & NTT_PRE_SCALE) && v) v = mont_mul(v, two_level(a.S.lo, a.S.hi, a.S.h, g, a.F), a.F);
            buf[g] = v;
        }
    }
    // stage s: butterflies of span half = n >> (s+1)
    static SMI_HD void stage(const SmallArgs &a, uint32_t s, uint32_t *buf, uint32_t tid) {
        const uint32_t n = 1u << a.L, hlog = a.L - s - 1, half = 1u << hlog;
        for (uint32_t u = tid; u < n / 2; u += SMI_NTT_THREADS) {
            const uint32_t blk = u >> hlog, pos = u & (half - 1);
            const uint32_t i = (blk << (hlog + 1)) + pos, j = i + half;
            const uint32_t x = buf[i], y = buf[j];
            buf[i] = fp_add(x, y, a.F.p);
            uint32_t d = fp_sub(x, y, a.F.p);
            const uint32_t e = pos << (a.T.K - (hlog + 1));  // w_{2half}^pos
            if (e) d = mont_mul(d, two_level(a.T.lo, a.T.hi, a.T.h, e, a.F), a.F);
            buf[j] = d;
        }
    }
    static SMI_HD void store(const SmallArgs &a, uint32_t batch, const uint32_t *buf, uint32_t tid) {
        const uint32_t n = 1u << a.L;
        uint32_t *out = a.out + (uint64_t)batch * a.out_stride;
        for (uint32_t k = tid; k < n; k += SMI_NTT_THREADS) {
            uint32_t loc = 0;
            for (uint32_t i = 0; i < a.L; i++) loc |= ((k >> i) & 1u) << (a.L - 1 - i);
            uint32_t v = buf[loc];
            if (a.flags & NTT_POST_SCALE) v = mont_mul(v, two_level(a.S.lo, a.S.hi, a.S.h, k, a.F), a.F);
            out[k] = v;
        }
    }
};
