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
    int logr[4];
};

// n = 2^L split into np digits of 6..10 bits, larger first.  Constraints the kernels rely on:
//   strided pass p: B_p >= W_p  <=>  L - S_{p+1} >= 12 - l_p
//   last pass:      R_0 >= W_last <=> l_0 >= 12 - l_last
inline bool ntt_plan_valid(uint32_t L, const NttPlan &pl) {
    if (pl.np < 2 || pl.np > 4) return false;
    int sum = 0;
    for (int i = 0; i < pl.np; i++) {
        if (pl.logr[i] < 6 || pl.logr[i] > 10) return false;
        sum += pl.logr[i];
    }
    if (sum != (int)L) return false;
    int consumed = 0;
    for (int i = 0; i + 1 < pl.np; i++) {
        consumed += pl.logr[i];
        if ((int)L - consumed < SMI_TILE_LOG - pl.logr[i]) return false;
    }
    return pl.logr[0] >= SMI_TILE_LOG - pl.logr[pl.np - 1];
}

inline NttPlan ntt_make_plan(uint32_t L) {
    NttPlan pl;
    memset(&pl, 0, sizeof pl);
    if (L <= SMI_TILE_LOG) return pl;  // np = 0
    // optional override for tuning: SMI_NTT_PLAN_<L>="10,10"
    char name[32];
    snprintf(name, sizeof name, "SMI_NTT_PLAN_%u", L);
    if (const char *env = getenv(name)) {
        NttPlan o;
        memset(&o, 0, sizeof o);
        const char *s = env;
        while (*s && o.np < 4) {
            o.logr[o.np++] = (int)strtol(s, (char **)&s, 10);
            if (*s == ',') s++;
        }
        if (ntt_plan_valid(L, o)) return o;
    }
    pl.np = L <= 20 ? 2 : (L <= 30 ? 3 : 4);
    int rem = (int)L;
    for (int i = 0; i < pl.np; i++) {
        int left = pl.np - i;
        pl.logr[i] = (rem + left - 1) / left;  // ceil: larger digits first
        rem -= pl.logr[i];
    }
    return pl;
}

// Geometric table entry (Montgomery form): c * q^(i*stride); q_m, c_m in Montgomery form.
SMI_HD uint32_t geom_entry(uint32_t c_m, uint32_t q_m, uint64_t i, uint64_t stride, const Fp &F) {
    return mont_mul(c_m, mont_pow(q_m, i * stride, F), F);
}
