// hash_quad.h -- Hash::combine (src/hash.rs:41-46) with one hash spread over a quad of lanes (device only).
//
// The upper levels of a Merkle tree have fewer nodes than a compute unit has lanes, so a level costs one
// node-hash *latency* (hash_core.h: ~1 310 dependent-ish instructions for a single lane).  Here lane q of an
// aligned quad holds words 4q..4q+3 of the paired-lane state (state bytes 4q..4q+3 and 16+4q..16+4q+3,
// i.e. the natural-layout digest words q and q+4): the S-box and the 4-byte linear mix are local to a
// lane (those four words are exactly linear-mix groups q and q+4), and the sequential ring add
// (src/hash.rs:77-81) becomes a local prefix plus a 4-lane exclusive scan over quad_perm DPP moves.
// Per mix and lane: 58 instead of 102 instructions.  The two 32-byte absorbs are computed redundantly by
// all four lanes (their byte recurrence is sequential anyway), each lane then keeps its two words.
// Same digests as hashc::node_hash: checked on the CPU by the emulator (csrc/emu.cpp runs this very code
// with `word` = a 4-lane vector stepping a quad in lockstep: SMI_QUAD_EMU), on the device by
// tools/quad_hash_test.hip and through the Merkle tests of the GPU suite.
#pragma once
#include "hash_core.h"

#if defined(__HIPCC__) || defined(SMI_QUAD_EMU)
namespace hashq {

#if !defined(SMI_QUAD_EMU)
#define SMI_QD __device__ __forceinline__
typedef uint32_t word;   // one lane's 32-bit value
// quad_perm DPP move: lane i of every aligned quad reads lane P_i of the same quad
template <int P0, int P1, int P2, int P3> SMI_QD word quad(word x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
#else
    return x;   // host pass of the compiler only: never executed
#endif
}
SMI_QD word lane_in_quad(uint32_t lane_id) { return lane_id & 3u; }
SMI_QD word mask_ge(word q, uint32_t k) { return q >= k ? ~0u : 0u; }
SMI_QD word mask_eq(word q, uint32_t k) { return q == k ? ~0u : 0u; }
SMI_QD word sel4(word q, word a, word b, word c, word d) { return q == 0 ? a : q == 1 ? b : q == 2 ? c : d; }
#endif
template <int I> SMI_QD word bcast(word x) { return quad<I, I, I, I>(x); }

// per-lane constants, set up once per kernel from q = lane & 3
struct Lane {
    word q;
    word rc[4], rc502[4];   // round-constant pairs of the lane's words 4q+k
    word ge1, ge2, ge3;     // all-ones where q >= 1, 2, 3
    word is0, is3;          // all-ones where q == 0, q == 3
};
SMI_QD Lane make_lane(uint32_t lane_id) {
    constexpr hashc::Consts C = hashc::make_consts();
    Lane L;
    L.q = lane_in_quad(lane_id);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        L.rc[k] = sel4(L.q, C.rc[k], C.rc[4 + k], C.rc[8 + k], C.rc[12 + k]);
        L.rc502[k] = sel4(L.q, C.rc502[k], C.rc502[4 + k], C.rc502[8 + k], C.rc502[12 + k]);
    }
    L.ge1 = mask_ge(L.q, 1);
    L.ge2 = mask_ge(L.q, 2);
    L.ge3 = mask_ge(L.q, 3);
    L.is0 = mask_eq(L.q, 0);
    L.is3 = mask_eq(L.q, 3);
    return L;
}

// src/hash.rs:59-86 on the lane's four words; PENDING as in hashc::mix_t (round constants of the
// previous mix folded into this S-box, its own left pending)
template <bool PENDING> SMI_QD void mix(word (&s)[4], const Lane &L) {
    using namespace hashc;
    const word kFE = vreg(0x00FE00FEu), kFF = vreg(0x00FF00FFu), k63 = vreg(0x00630063u), kHI = vreg(0xFFFF0000u);
    word r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const word t = pk_mad_u16(s[k], word(0x01F601F6u), PENDING ? L.rc502[k] : word(0u));
        r[k] = bfi32(kFE, t, t >> 8);
    }
    const word T = xor3(xor3(r[0], r[1], r[2]), r[3], k63);
    const word s0 = (T ^ r[2]) & kFF, s1 = (T ^ r[1]) & kFF, s2 = (T ^ r[3]) & kFF, s3 = (T ^ r[0]) & kFF;
    // ring add, in the notation of hashc::mix_t (word w = 4q+k): tot = sum of all 16 words (lane 0 of the
    // word: A), S0/S1/S15 = words 0, 1, 15, N0 = new word 0; every lane computes N0
    word tot = add3(s0, s1, s2) + s3;
    tot = tot + quad<1, 0, 3, 2>(tot);
    tot = tot + quad<2, 3, 0, 1>(tot);
    const word S0 = bcast<0>(s0), S1 = bcast<0>(s1), S15 = bcast<3>(s3);
    const word m0 = S0 * 0xFFFF0001u;
    const word N0 = add3((tot << 17) + m0, S1, dup_hi16(S15)) + (S0 & kHI);
    // d_k = what word 4q+k adds to the running value: s[w] + s[w+1], except word 0 (N0 itself) and
    // word 15 (s[15] + (b_0 | new[0] << 16))
    const word nxt = quad<1, 2, 3, 0>(s0);
    const word d0 = bfi32(L.is0, N0, s0 + s1), d1 = s1 + s2, d2 = s2 + s3;
    const word d3 = s3 + bfi32(L.is3, funnel16(N0, S0), nxt);
    const word p0 = d0, p1 = p0 + d1, p2 = p1 + d2, p3 = p2 + d3;
    // exclusive scan of the lane totals
    const word E = (bcast<0>(p3) & L.ge1) + (bcast<1>(p3) & L.ge2) + (bcast<2>(p3) & L.ge3);
    s[0] = p0 + E;
    s[1] = p1 + E;
    s[2] = p2 + E;
    s[3] = p3 + E;
}
SMI_QD void flush(word (&s)[4], const Lane &L) {
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = s[k] + L.rc[k];
}

// natural words q and q+4 of P -> the lane's four state words, and back (low byte of each lane only)
SMI_QD void from_words(const word P[8], word (&s)[4], const Lane &L) {
    const word lo = sel4(L.q, P[0], P[1], P[2], P[3]);
    const word hi = sel4(L.q, P[4], P[5], P[6], P[7]);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) s[k] = hashc::perm8(hi, lo, 0x0C000C00u | ((4u + k) << 16) | k);
}
SMI_QD void to_words(const word (&s)[4], word &lo, word &hi) {
    using hashc::perm8;
    lo = perm8(s[1], s[0], 0x0C0C0400u) | perm8(s[3], s[2], 0x04000C0Cu);
    hi = perm8(s[1], s[0], 0x0C0C0602u) | perm8(s[3], s[2], 0x06020C0Cu);
}

// Hash::combine over a quad: every lane passes the same l and r; lane q returns digest words q (lo)
// and q+4 (hi).  All four lanes of the quad must be active.
SMI_QD void node_hash(const word l[8], const word r[8], const Lane &L, word &lo, word &hi) {
    constexpr hashc::InitWords I = hashc::make_init_words();
    word P[8], s[4];
#pragma unroll
    for (int j = 0; j < 8; j++) P[j] = I.p[j];
    hashc::absorb32_words(P, l);
    from_words(P, s, L);
    mix<false>(s, L);
    flush(s, L);
    to_words(s, lo, hi);
    P[0] = bcast<0>(lo); P[1] = bcast<1>(lo); P[2] = bcast<2>(lo); P[3] = bcast<3>(lo);
    P[4] = bcast<0>(hi); P[5] = bcast<1>(hi); P[6] = bcast<2>(hi); P[7] = bcast<3>(hi);
    hashc::absorb32_words(P, r);
    from_words(P, s, L);
    mix<false>(s, L);
#pragma unroll
    for (int k = 0; k < 8; k++) mix<true>(s, L);
    flush(s, L);
    to_words(s, lo, hi);
}

}  // namespace hashq
#endif
