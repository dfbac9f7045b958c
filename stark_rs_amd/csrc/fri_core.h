// fri_core.h -- the per-element arithmetic of Fri::fold_codeword (reference src/fri.rs:57-91),
// shared by the HIP kernel and the CPU emulator of the non-GPU tests.
#pragma once
#include "ntt_core.h"

// out = 2^-1 * ((1 + a/x_i) lo + (1 - a/x_i) hi)  =  2^-1 (lo + hi) + (a/2 * x_i^-1) (lo - hi)
// with x_i^-1 = offset^-1 * omega^-i taken from the two-level table S at the GLOBAL index i
// (a shard of a distributed codeword passes its own index range), ah_m = alpha/2 and inv2_m =
// 2^-1 in Montgomery form.
SMI_HD uint32_t fold_element(uint32_t lo, uint32_t hi, uint32_t i, uint32_t ah_m, uint32_t inv2_m, const ScaleTables &S, const Fp &F) {
    const uint32_t s = fp_add(lo, hi, F.p), d = fp_sub(lo, hi, F.p);
    const uint32_t t_m = mont_mul(two_level(S.lo, S.hi, S.h, i, F), ah_m, F);
    return fp_add(mont_mul(s, inv2_m, F), mont_mul(d, t_m, F), F.p);
}
SMI_HD uint32_t fold_alpha_half(uint64_t alpha, uint32_t inv2_m, const Fp &F) {
    return mont_mul(to_mont_u64(alpha, F), inv2_m, F);   // alpha may be an unreduced u64 (H6); no 64-bit division per thread
}
