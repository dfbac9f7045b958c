// tables.h -- specifications of the twiddle / scale tables (shared by ntt.hip and emu.cpp).
// Every table is a geometric sequence c * q^(i*stride) in Montgomery form, generated on
// the device by geom_table_kernel (host loop in the emulator).
#pragma once
#include "ntt_host.h"

struct GeomSpec {
    uint32_t c_m, q_m;  // Montgomery form
    uint64_t stride;
    uint32_t count;
    uint32_t pair;      // 1: write Tw2 entries (plain value, Shoup quotient floor(value * 2^32 / p))
};

struct FieldSetup {
    Fp F;
    uint32_t g;          // generator (plain)
    uint32_t K;          // usable two-adicity (log2 of the largest transform), capped at 27
    uint32_t wmax[2];    // primitive 2^K-th root: [0] forward, [1] inverse (plain form)
};

// Returns false when p is unusable (composite, even, >= 2^30 -- the lazy butterflies need 4p < 2^32 --,
// two-adicity < 12, or g not of full 2-power order).
// Deterministic Miller-Rabin for n < 2^32 (witnesses 2, 7, 61).
inline bool is_prime_u32(uint32_t n) {
    if (n < 2) return false;
    for (uint32_t q : {2u, 3u, 5u, 7u, 11u, 13u}) {
        if (n == q) return true;
        if (n % q == 0) return false;
    }
    uint32_t d = n - 1, r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (uint32_t a : {2u, 7u, 61u}) {
        uint32_t x = host_powmod(a % n, d, n);
        if (x == 1 || x == n - 1 || a % n == 0) continue;
        bool witness = true;
        for (uint32_t i = 1; i < r && witness; i++) {
            x = host_mulmod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

inline bool field_setup(uint64_t p64, uint64_t g64, FieldSetup *fs) {
    if (p64 < 3 || p64 >= (1ull << 30) || (p64 & 1) == 0 || g64 == 0 || g64 >= p64) return false;
    const uint32_t p = (uint32_t)p64, g = (uint32_t)g64;
    if (!is_prime_u32(p)) return false;   // every inverse here is a Fermat power: the modulus must be prime
    uint32_t two_adicity = 0;
    while (((p - 1) >> two_adicity) % 2 == 0) two_adicity++;
    if (two_adicity < SMI_TILE_LOG) return false;
    fs->F = fp_make(p);
    fs->g = g;
    fs->K = two_adicity > 27 ? 27 : two_adicity;
    const uint32_t w = host_powmod(g, (p - 1) >> fs->K, p);
    if (host_powmod(w, 1ull << (fs->K - 1), p) != p - 1) return false;  // w must have order exactly 2^K
    fs->wmax[0] = w;
    fs->wmax[1] = host_powmod(w, p - 2, p);
    return true;
}

inline uint32_t ntt_table_h(uint32_t K) { return (K + 1) / 2; }

// specs[0] = tw10, specs[1] = lo, specs[2] = hi for one direction
inline void ntt_table_specs(const FieldSetup &fs, int inverse, GeomSpec specs[3]) {
    const Fp &F = fs.F;
    const uint32_t w_m = (uint32_t)(((uint64_t)fs.wmax[inverse ? 1 : 0] << 32) % F.p);
    const uint32_t h = ntt_table_h(fs.K);
    specs[0] = GeomSpec{F.r1, w_m, 1ull << (fs.K - SMI_TW_LOG), 1u << SMI_TW_LOG, 1};
    specs[1] = GeomSpec{F.r1, w_m, 1, 1u << h, 0};
    specs[2] = GeomSpec{F.r1, w_m, 1ull << h, 1u << (fs.K - h), 0};
}

// c * q^i for i < 2^L, two-level: lo (2^h entries, ratio q), hi (2^(L-h) entries, c * q^(i<<h))
inline uint32_t scale_table_h(uint32_t L) { return (L + 1) / 2; }
inline void scale_table_specs(const Fp &F, uint32_t c_plain, uint32_t q_plain, uint32_t L, GeomSpec specs[2]) {
    const uint32_t c_m = (uint32_t)(((uint64_t)c_plain << 32) % F.p), q_m = (uint32_t)(((uint64_t)q_plain << 32) % F.p);
    const uint32_t h = scale_table_h(L);
    specs[0] = GeomSpec{F.r1, q_m, 1, 1u << h, 0};
    specs[1] = GeomSpec{c_m, q_m, 1ull << h, 1u << (L - h), 0};
}

// Montgomery-form value -> plain value + Shoup quotient (64-bit divide: table generation only)
SMI_HD Tw2 tw2_from_mont(uint32_t v_m, const Fp &F) {
    const uint32_t w = from_mont(v_m, F);
    return Tw2{w, (uint32_t)(((uint64_t)w << 32) / F.p)};
}
