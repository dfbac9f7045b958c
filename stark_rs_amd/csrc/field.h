// field.h -- prime-field arithmetic for p < 2^31, u32 storage.
//
// Device counterpart of FiniteField::{add,sub,mul,neg} (reference src/ff.rs:138-167).
// The reference widens to u128 and takes `% p`; every result is the canonical residue
// in [0,p), so any exact modular arithmetic is bit-identical.  Here: conditional-subtract
// add/sub and a 32x32->64 Montgomery product (R = 2^32).  Twiddles are stored in
// Montgomery form (w*R mod p) so mont_mul(x, wR) = x*w mod p keeps data in plain form.
//
// Functions are host+device so the index/arith logic can also be exercised by the
// host-side emulator used in CPU tests (no GPU in the build container).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SMI_HD __host__ __device__ __forceinline__
#else
#define SMI_HD inline
#endif

struct Fp {
    uint32_t p;     // modulus, odd, < 2^31
    uint32_t pinv;  // p^-1 mod 2^32
    uint32_t r1;    // R mod p      (Montgomery form of 1)
    uint32_t r2;    // R^2 mod p    (to_mont(x) = mont_mul(x, r2))
};

SMI_HD uint32_t fp_add(uint32_t a, uint32_t b, uint32_t p) {
    uint32_t s = a + b;  // a,b < p < 2^31: no wrap
    return s >= p ? s - p : s;
}
SMI_HD uint32_t fp_sub(uint32_t a, uint32_t b, uint32_t p) {
    const uint32_t d = a - b;   // wraps iff a < b, and then d + p wraps back into [0,p)
    return d < d + p ? d : d + p;
}
SMI_HD uint32_t fp_neg(uint32_t a, uint32_t p) { return a ? p - a : 0; }

SMI_HD uint32_t umulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
SMI_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
SMI_HD uint32_t bitrev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = (x >> 16) | (x << 16);
    x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
    return ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
#endif
}

// A value held in a VGPR.  On gfx950 the simple VALU ops (add, sub, and, or, xor, right shifts,
// v_bitop3) issue in 2 cycles per wave only while every source is a VGPR or an inline constant;
// with an SGPR or literal source they take 4, like the multiplies and the other three-operand
// ops (measured, tools/ubench_valu.hip).  The empty asm is pure: it can be hoisted out of loops.
SMI_HD uint32_t vreg(uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SMI_NO_VREG)   // SMI_NO_VREG: tuning builds only
    asm("" : "+v"(c));
#endif
    return c;
}

// a*b*R^-1 mod p for a*b < p*2^32 (e.g. a < p, b arbitrary u32).  Result in [0,p).
// t = a*b, m = lo(t)*p^-1 mod 2^32; t - m*p is a multiple of 2^32 and (t - m*p)/2^32 =
// hi(t) - hi(m*p) lies in (-p, p) -- the low words cancel exactly.  v_mul_lo/hi_u32 issue at the
// full integer VALU rate on gfx950 (measured, tools/ubench_valu.hip) while v_mad_u64_u32 costs
// ~2.5 slots, so the 64-bit product is spelled as a lo/hi pair; the final correction is the
// 3-op min(r, r+p) form (r+p wraps back into [0,p) exactly when r is negative).
SMI_HD uint32_t mont_mul(uint32_t a, uint32_t b, const Fp &F) {
    const uint32_t lo = a * b, hi = umulhi32(a, b);
    const uint32_t u = umulhi32(lo * F.pinv, F.p);
    const uint32_t r = hi - u;
    return umin32(r, r + F.p);
}
// Same product for a multiplicand b that comes with its companion bq = b * p^-1 mod 2^32
// (twiddle tables store the pair): one multiply fewer on the dependent chain.
SMI_HD uint32_t mont_mul_c(uint32_t a, uint32_t b, uint32_t bq, const Fp &F) {
    const uint32_t hi = umulhi32(a, b);
    const uint32_t u = umulhi32(a * bq, F.p);
    const uint32_t r = hi - u;
    return umin32(r, r + F.p);
}
SMI_HD uint32_t to_mont(uint32_t a, const Fp &F) { return mont_mul(a, F.r2, F); }
SMI_HD uint32_t from_mont(uint32_t a, const Fp &F) { return mont_mul(a, 1u, F); }

// Montgomery form of an arbitrary u64 reduced mod p (an unreduced Fiat-Shamir challenge, src/fiat_shamir.rs:23-24) without a
// 64-bit division: v = hi * 2^32 + lo, R = 2^32, so v * R = (hi * R) * R + lo * R (mod p) -- three Montgomery products, each
// with R^2 < p as one factor and an arbitrary u32 as the other.  Equals to_mont(v % p).
SMI_HD uint32_t to_mont_u64(uint64_t v, const Fp &F) {
    const uint32_t hi_m = mont_mul(mont_mul((uint32_t)(v >> 32), F.r2, F), F.r2, F);
    return fp_add(hi_m, mont_mul((uint32_t)v, F.r2, F), F.p);
}

// base^e with base in Montgomery form; result in Montgomery form.
SMI_HD uint32_t mont_pow(uint32_t base_m, uint64_t e, const Fp &F) {
    uint32_t res = F.r1;
    while (e) {
        if (e & 1) res = mont_mul(res, base_m, F);
        base_m = mont_mul(base_m, base_m, F);
        e >>= 1;
    }
    return res;
}

// Reduce an arbitrary u64 (e.g. an unreduced Fiat-Shamir challenge, reference
// src/fiat_shamir.rs:23-24) to [0,p).
SMI_HD uint32_t reduce_u64(uint64_t v, uint32_t p) { return (uint32_t)(v % p); }

// ---- host-only helpers -------------------------------------------------------------
inline uint32_t host_mulmod(uint32_t a, uint32_t b, uint32_t p) { return (uint32_t)(((uint64_t)a * b) % p); }
inline uint32_t host_powmod(uint32_t b, uint64_t e, uint32_t p) {
    uint32_t r = 1 % p;
    while (e) {
        if (e & 1) r = host_mulmod(r, b, p);
        b = host_mulmod(b, b, p);
        e >>= 1;
    }
    return r;
}
inline Fp fp_make(uint32_t p) {
    Fp F;
    F.p = p;
    uint32_t inv = p;  // Newton: inv*p == 1 mod 2^32 (p odd; 3 correct bits to start)
    for (int i = 0; i < 5; i++) inv *= 2u - p * inv;
    F.pinv = inv;
    F.r1 = (uint32_t)((1ull << 32) % p);
    F.r2 = (uint32_t)(((uint64_t)F.r1 * F.r1) % p);
    return F;
}
