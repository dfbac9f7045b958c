// ntt_host.h -- host-side planning shared by the HIP driver (ntt.hip) and the CPU
// emulator used by the non-GPU tests (emu.cpp).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ntt_core.h"

struct NttPlan {
    int np;        // 0 = single small-kernel launch
    int logr[4];   // digit of pass p
    int logw[4];   // lines per tile of pass p (tile = 2^(logr+logw) points, 12..14)
};

// Kernels exist for these (logr, logw) pairs (ntt.hip instantiates exactly this list).
#define SMI_NTT_SHAPES(X) \
    X(6, 6) X(7, 5) X(8, 4) X(9, 3) X(10, 2) /* 4096-point tiles  */ \
    X(7, 6) X(8, 5) X(9, 4) X(10, 3)         /* 8192-point tiles  */ \
    X(8, 6) X(9, 5) X(10, 4) X(11, 3)        /* 16384-point tiles */ \
    X(11, 2)                                 /* 8192-point tile, 2048-point lines */
inline bool ntt_shape_ok(int lr, int lw) {
#define X(a, b) if (lr == a && lw == b) return true;
    SMI_NTT_SHAPES(X)
#undef X
    return false;
}

// n = 2^L split into np digits.  Constraints the kernels rely on:
//   strided pass p: B_p >= W_p   <=>  L - S_{p+1} >= logw_p
//   last pass:      R_0 >= W_last <=> logr_0 >= logw_last
inline bool ntt_plan_valid(uint32_t L, const NttPlan &pl) {
    if (pl.np < 2 || pl.np > 4) return false;
    int sum = 0;
    for (int i = 0; i < pl.np; i++) {
        if (!ntt_shape_ok(pl.logr[i], pl.logw[i])) return false;
        sum += pl.logr[i];
    }
    if (sum != (int)L) return false;
    int consumed = 0;
    for (int i = 0; i + 1 < pl.np; i++) {
        consumed += pl.logr[i];
        if ((int)L - consumed < pl.logw[i]) return false;
    }
    return pl.logr[0] >= pl.logw[pl.np - 1];
}

inline NttPlan ntt_make_plan(uint32_t L, uint32_t batch = 1) {
    NttPlan pl;
    memset(&pl, 0, sizeof pl);
    // np = 0: the single-workgroup kernel; a large batch of 4096-point columns is better served by two
    // passes of 64-point lines (measured: 4096 columns 112 -> see DESIGN.md)
    if (L < SMI_TILE_LOG || (L == SMI_TILE_LOG && batch < 64)) return pl;
    // optional override for tuning: SMI_NTT_PLAN_<L>="10.2,10.2" (logr.logw per pass)
    char name[32];
    snprintf(name, sizeof name, "SMI_NTT_PLAN_%u", L);
    if (const char *env = getenv(name)) {
        NttPlan o;
        memset(&o, 0, sizeof o);
        const char *s = env;
        while (*s && o.np < 4) {
            o.logr[o.np] = (int)strtol(s, (char **)&s, 10);
            if (*s == '.') s++;
            o.logw[o.np] = (int)strtol(s, (char **)&s, 10);
            o.np++;
            if (*s == ',') s++;
        }
        if (ntt_plan_valid(L, o)) return o;
    }
    // 2^21..2^22: two passes of 2048-point lines only pay off for a single column (fewer launches);
    // batched columns run faster as three lighter passes (measured)
    pl.np = (L <= 20 || (L <= 22 && batch == 1)) ? 2 : (L <= 30 ? 3 : 4);
    int rem = (int)L;
    for (int i = 0; i < pl.np; i++) {
        int left = pl.np - i;
        pl.logr[i] = (rem + left - 1) / left;  // ceil: larger digits first
        rem -= pl.logr[i];
    }
    // 2^23..2^26 in three passes: 256-point columns in 64-column tiles first (256-byte store runs), 512-point lines last, what
    // is left in the middle (with 64-column tiles while its lines are short).  Measured in the sustained regime at the end of
    // r03 (profiles/r03_plans5_ab.log, r03_plans6_sizes.log; the earlier sweeps never paired a 16 K-point last tile with a
    // 64-column first one): 2^25 x 4 extension step 0.698-0.702 -> 0.675-0.678 ms, 2^23 x 4 0.169 -> 0.163, one 2^26-point
    // transform 0.414-0.421 -> 0.390 ms, 2^24 x 4 within 0.5 % of the even split.
    if (pl.np == 3 && L >= 23 && L <= 26) {
        NttPlan o = pl;
        const int mid = (int)L - 17;
        o.logr[0] = 8; o.logw[0] = 6;
        o.logr[1] = mid; o.logw[1] = mid <= 7 ? 6 : 5;
        o.logr[2] = 9; o.logw[2] = 5;
        if (ntt_plan_valid(L, o)) return o;
    }
    // widest lines that fit: strided passes want >= 128-byte runs (W >= 32), see DESIGN.md
    int consumed = 0;
    for (int i = 0; i < pl.np; i++) {
        consumed += pl.logr[i];
        const bool last = i == pl.np - 1;
        const int limit = last ? pl.logr[0] : (int)L - consumed;
        int best = -1;
        for (int lw = 2; lw <= 6; lw++) {
            if (!ntt_shape_ok(pl.logr[i], lw) || lw > limit || pl.logr[i] + lw > (last ? 13 : 14) || lw > 5) continue;
            // small problems: keep at least ~2 workgroups per CU in flight rather than wide lines
            const uint64_t tiles = ((uint64_t)batch << L) >> (pl.logr[i] + lw);
            if (best >= 0 && tiles < 512) continue;
            best = lw;
        }
        if (best < 0)
            for (int lw = 6; lw >= 2; lw--)
                if (ntt_shape_ok(pl.logr[i], lw) && lw <= limit) best = lw;
        pl.logw[i] = best;
    }
    return pl;
}

// can one 2^L-point transform be sharded over 2^log_g ranks on the pass pipeline (ntt_driver.h)?
inline bool ntt_shard_ok(uint32_t L, uint32_t log_g) {
    const NttPlan pl = ntt_make_plan(L, 1);
    if (pl.np < 2) return false;
    const int blog = (int)L - pl.logr[0];
    return blog - (int)log_g >= pl.logw[0] && pl.logr[0] - (int)log_g >= pl.logw[pl.np - 1] && pl.logr[0] >= (int)log_g;
}

// log2 of the radix of the last in-tile step for a digit of logr bits (Steps<logr> in ntt_core.h)
inline int ntt_last_step_log(int logr) { return logr == 6 ? 2 : logr == 7 ? 3 : logr == 8 ? 4 : logr == 9 ? 2 : 3; }  // 10, 11 -> 3

// Geometric table entry (Montgomery form): c * q^(i*stride); q_m, c_m in Montgomery form.
SMI_HD uint32_t geom_entry(uint32_t c_m, uint32_t q_m, uint64_t i, uint64_t stride, const Fp &F) {
    return mont_mul(c_m, mont_pow(q_m, i * stride, F), F);
}
