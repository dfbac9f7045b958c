// hash_core.h -- the reference's 32-byte hash (src/hash.rs:7-99) for one lane (host+device).
//
// State layout ("paired lanes"): 16 u32 words, word w holds state byte w in bits 0..7 and
// state byte w+16 in bits 16..23; bits 8..15 / 24..31 are headroom.  (State2, further down, puts
// byte w of two different hashes in the two lanes instead: what the Merkle kernels use.)  With it
//   * the byte S-box  rotl1(b*251)^0x63  is one packed 16-bit multiply-add by 502 on two bytes at
//     once (bit 8 of 502*b is the rotated-out bit), src/hash.rs:88-94;
//   * the 4-byte linear mix (src/hash.rs:64-75) is plain XORs of whole words: groups g and
//     g+4 share words 4g..4g+3;
//   * the sequential in-place ring add (src/hash.rs:77-81) runs on both lanes at once, one
//     three-input add per word, once the two lane crossings have been resolved from the byte
//     sum of lane 0 (see mix_t); lane sums stay < 2^16, so the two bytes of a word never
//     interfere.  The round constants (src/hash.rs:83-85,96-99) ride in the next S-box's addend.
// Everything is integer/byte work: VALU-bound, no LDS, no MFMA.
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "field.h"
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define SMI_MUL24(a, b) __umul24((a), (b))
#else
#define SMI_MUL24(a, b) (((uint32_t)(a) & 0xFFFFFFu) * ((uint32_t)(b) & 0xFFFFFFu))
#endif

namespace hashc {

// src/hash.rs:53 and :96-99
#define SMI_PRIMES {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53}
#define SMI_RC                                                                                          \
    {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36, 0x6c, 0xd8, 0xab, 0x4d, 0x9a, 0x2f,    \
     0x5e, 0xbc, 0x63, 0xc6, 0x97, 0x35, 0x6a, 0xd4, 0xb3, 0x7d, 0xfa, 0xef, 0xc5, 0x91, 0x39, 0x72}

struct Consts {
    uint32_t rc[16];     // round-constant pair of word w: RC[w] | RC[w+16] << 16
    uint32_t rc502[16];  // 502 * RC per 16-bit lane (mod 2^16): the S-box addend that applies a pending RC
};
constexpr Consts make_consts() {
    const uint8_t rc[32] = SMI_RC;
    Consts c{};
    for (int w = 0; w < 16; w++) {
        c.rc[w] = (uint32_t)rc[w] | ((uint32_t)rc[w + 16] << 16);
        c.rc502[w] = ((502u * rc[w]) & 0xFFFFu) | (((502u * rc[w + 16]) & 0xFFFFu) << 16);
    }
    return c;
}

// a*m + c on the two 16-bit lanes independently (v_pk_mad_u16): dirt above bit 7 of a lane never
// reaches the other lane, and the low 9 bits of lane*502 depend only on the lane's low 8 bits.
SMI_HD uint32_t pk_mad_u16(uint32_t a, uint32_t m, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 r = __builtin_bit_cast(us2, a) * __builtin_bit_cast(us2, m) + __builtin_bit_cast(us2, c);
    return __builtin_bit_cast(uint32_t, r);
#else
    const uint32_t lo = ((a & 0xFFFFu) * (m & 0xFFFFu) + (c & 0xFFFFu)) & 0xFFFFu;
    const uint32_t hi = ((a >> 16) * (m >> 16) + (c >> 16)) & 0xFFFFu;
    return lo | (hi << 16);
#endif
}

// Three-input bit operations as single VALU ops (gfx950 v_bitop3_b32; result bit = TT[(a << 2) | (b << 1) | c]).
// Through the compiler's builtin rather than inline asm: the optimizer folds them on constants, schedules them
// like any other VALU op and needs no hazard s_nop after them (an asm result read by the next instruction costs
// one: 130 of them per thread in merkle_sub_kernel, r03).  It issues at twice the rate of v_bfi_b32 / v_xor3
// (tools/ubench_valu.hip).
template <int TT> SMI_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (int i = 0; i < 8; i++)
        if ((TT >> i) & 1) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
#endif
}
SMI_HD uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) { return bitop3<0xCA>(mask, a, b); }   // (mask & a) | (~mask & b)
SMI_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return bitop3<0x96>(a, b, c); }
// Where every input of an S-box tail or of a linear-mix XOR is a compile-time constant (the half of a leaf's
// state the 8-byte message never reaches, in its first mix) these wrappers take the plain C expression, which
// folds away (a leftover of the inline-asm forms; harmless with the builtin).
#if defined(__HIP_DEVICE_COMPILE__)
#define SMI_CONST_P(x) __builtin_constant_p(x)
#else
#define SMI_CONST_P(x) 0
#endif
SMI_HD uint32_t sbox_tail(uint32_t t, uint32_t kFE) {          // (t & 0xFE) | (t >> 8 & 1) per lane (dirt above bit 7 stays)
    if (SMI_CONST_P(t)) return (t & 0x00FE00FEu) | ((t >> 8) & 0xFF01FF01u);
    return bfi32(kFE, t, t >> 8);
}
SMI_HD uint32_t lin_sum(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t k63) {   // t0^t1^t2^t3 ^ 0x63 per lane
    if (SMI_CONST_P(t0) && SMI_CONST_P(t1) && SMI_CONST_P(t2) && SMI_CONST_P(t3)) return t0 ^ t1 ^ t2 ^ t3 ^ 0x00630063u;
    return xor3(xor3(t0, t1, t2), t3, k63);
}
SMI_HD uint32_t lin_out(uint32_t T, uint32_t t, uint32_t kFF) {   // (T ^ t), lanes masked clean
    if (SMI_CONST_P(T) && SMI_CONST_P(t)) return (T ^ t) & 0x00FF00FFu;
    return (T ^ t) & kFF;
}

// v_add3_u32 costs what two v_add_u32 cost (4 cycles per wave either way), so the choice is left
// to the compiler: it keeps s[w] + s[w+1] off the dependent chain of the ring add.
SMI_HD uint32_t add3(uint32_t a, uint32_t b, uint32_t c) { return a + (b + c); }
// a + b that stays an add of its own (the empty asm hides it from the pattern that forms v_add3_u32); used by
// tools/ubench_mix.hip's two-add variant of the ring add (measured and not adopted, see mix2_t)
SMI_HD uint32_t pair_sum(uint32_t a, uint32_t b) {
    uint32_t r = a + b;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(r));
#endif
    return r;
}

// The high 16-bit lane copied into both lanes (v_perm_b32), and {hi:lo} >> 16 (v_alignbit_b32).
SMI_HD uint32_t dup_hi16(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(a, a, 0x03020302u);
#else
    return (a >> 16) * 0x00010001u;
#endif
}
SMI_HD uint32_t funnel16(uint32_t hi, uint32_t lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, 16);
#else
    return (lo >> 16) | (hi << 16);
#endif
}

// {hi:lo} >> 8 bits*n for n in 1..3 (v_alignbit_b32), and the byte permute of {hi:lo}
// (v_perm_b32): selector byte 0..3 picks a byte of lo, 4..7 a byte of hi, 0x0C gives 0x00.
SMI_HD uint32_t funnel(uint32_t hi, uint32_t lo, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, bits);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> bits);
#endif
}
SMI_HD uint32_t perm8(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t c = (sel >> (8 * k)) & 0xFFu;
        const uint32_t b = c < 8 ? (uint32_t)(v >> (8 * c)) & 0xFFu : (c == 0x0Cu ? 0u : 0xFFu);
        r |= b << (8 * k);
    }
    return r;
#endif
}

// The VGPR-held constants of a mix (masks, the deferred ^0x63, the S-box multiplier): made ONCE per hash by the
// caller and passed down, so that the closing-mix loops do not re-materialise them every iteration (the empty asm
// of vreg() is re-executed wherever its call is inlined: 4 v_mov per mix2 inside the loops before, r03).
struct MixK {
    uint32_t kFE, kFF, k63, k502, kHI;
};
SMI_HD MixK mix_consts() {
    return MixK{vreg(0x00FE00FEu), vreg(0x00FF00FFu), vreg(0x00630063u), vreg(0x01F601F6u), vreg(0xFFFF0000u)};
}

// State convention: the TRUE state byte is (stored lane + pending round constant) mod 256, where
// the round constants of the previous mix (src/hash.rs:83-85) may still be pending; only the low 8
// bits of a lane are meaningful (bits 8..15 may hold carry dirt below 2^16).
struct State {
    uint32_t s[16];
};

SMI_HD void init(State &st) {
    const uint8_t pr[16] = SMI_PRIMES;
#pragma unroll
    for (int w = 0; w < 16; w++) st.s[w] = (uint32_t)pr[w] * 0x00010001u;  // bytes w and w+16 are both PRIMES[w]
}

// Apply the pending round constants (needed before absorbing or reading the digest).
SMI_HD void flush(State &st) {
    constexpr Consts C = make_consts();
#pragma unroll
    for (int w = 0; w < 16; w++) st.s[w] += C.rc[w];
}

// src/hash.rs:59-86.  PENDING: the previous mix's round constants have not been added yet (they
// are folded into this S-box's multiply-add).  Leaves its own round constants pending.
template <bool PENDING> SMI_HD void mix_t(State &st, const MixK &K) {
    constexpr Consts C = make_consts();
    uint32_t *s = st.s;
    // (1) S-box  rotl1(251*b) (^0x63 deferred): t = 502*(b + rc); result = (t & 0xFE) | bit 8 of t.
    // Bits 8..15 of the result lanes are left dirty; the linear layer's last XOR masks them.
    const uint32_t kFE = K.kFE, kFF = K.kFF, k63 = K.k63, kHI = K.kHI, k502 = K.k502;   // k502: see mix2_t
    uint32_t r[16];
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t t = pk_mad_u16(s[w], SMI_CONST_P(s[w]) ? 0x01F601F6u : k502, PENDING ? C.rc502[w] : 0u);
        r[w] = sbox_tail(t, kFE);
    }
    // (2) linear mix: new = (t0^t1^t2^t3) ^ {t2, t1, t3, t0}, with the deferred ^0x63 (it passes
    // through the three-byte XORs unchanged); the final XOR also masks the lanes clean.
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t0 = r[4 * q], t1 = r[4 * q + 1], t2 = r[4 * q + 2], t3 = r[4 * q + 3];
        const uint32_t T = lin_sum(t0, t1, t2, t3, k63);
        s[4 * q] = lin_out(T, t2, kFF);
        s[4 * q + 1] = lin_out(T, t1, kFF);
        s[4 * q + 2] = lin_out(T, t3, kFF);
        s[4 * q + 3] = lin_out(T, t0, kFF);
    }
    // (3) ring add (src/hash.rs:77-81), round constants stay pending.  With a_w = byte w (lane 0) and
    // b_w = byte 16+w (lane 1) the sequential in-place recurrence is lane-parallel for words 1..14:
    //     N[w] = N[w-1] + s[w] + s[w+1]                                  (one v_add3 per word)
    // and the two places where the lanes meet are closed up front from A = sum_w a_w:
    //     new[15] = a_15 + b_0 + new[14] = 2A - a_0 + b_0 + b_15   (unrolling new[1..14])
    //     N[0]  = (a_0 + a_1 + b_15) | (b_0 + b_1 + new[15]) << 16
    //     N[15] = N[14] + s[15] + (b_0 | new[0] << 16)
    // Lane 0 never exceeds 16*2*255 + 765 < 2^16, so it cannot carry into lane 1; lane 1 may wrap
    // out of the top of the word (only the low 8 bits of a lane are meaningful afterwards).
    const uint32_t t1 = add3(s[0], s[1], s[2]), t2 = add3(s[3], s[4], s[5]), t3 = add3(s[6], s[7], s[8]);
    const uint32_t t4 = add3(s[9], s[10], s[11]), t5 = add3(s[12], s[13], s[14]);
    const uint32_t tot = add3(t1, t2, t3) + add3(t4, t5, s[15]);   // lane 0: A (< 4096)
    const uint32_t m0 = s[0] * 0xFFFF0001u;                         // lane 0: a_0, lane 1: b_0 - a_0
    uint32_t N[16];
    N[0] = add3((tot << 17) + m0, s[1], dup_hi16(s[15])) + (s[0] & kHI);
#pragma unroll
    for (int w = 1; w < 15; w++) N[w] = add3(N[w - 1], s[w], s[w + 1]);
    N[15] = add3(N[14], s[15], funnel16(N[0], s[0]));               // + (b_0 | new[0] << 16)
#pragma unroll
    for (int w = 0; w < 16; w++) s[w] = N[w];
}

template <bool PENDING> SMI_HD void mix_t(State &st) { mix_t<PENDING>(st, mix_consts()); }
// One complete mix_state on a fully applied state (used by the single-lane transcript kernels).
SMI_HD void mix(State &st, const MixK &K) {
    mix_t<false>(st, K);
    flush(st);
}
SMI_HD void mix(State &st) { mix(st, mix_consts()); }

// byte accessors in the paired-lane layout
SMI_HD uint32_t get_byte(const State &st, int i) { return (st.s[i & 15] >> ((i & 16) ? 16 : 0)) & 0xFFu; }
SMI_HD void xor_byte(State &st, int i, uint32_t v) { st.s[i & 15] ^= v << ((i & 16) ? 16 : 0); }
SMI_HD void set_byte(State &st, int i, uint32_t v) {
    const int sh = (i & 16) ? 16 : 0;
    st.s[i & 15] = (st.s[i & 15] & ~(0xFFu << sh)) | (v << sh);
}

// src/hash.rs:15-20 for one byte at position pos (compile-time after unrolling)
SMI_HD void absorb_byte(State &st, int pos, uint32_t byte) {
    uint32_t v = (get_byte(st, pos) + byte) & 0xFFu;
    v = ((v << 3) | (v >> 5)) & 0xFFu;
    set_byte(st, pos, v);
    xor_byte(st, (pos + 7) & 31, v);
}

// ---- "natural" layout: 8 little-endian words, bytes 4j..4j+3 in P[j] -- the layout of a digest and
// of a message chunk.  Whole 32-byte chunks are absorbed in it, four bytes per op.

// paired lanes -> natural; only the low byte of each lane is read, so carry dirt is ignored
SMI_HD void to_words(const State &st, uint32_t d[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int w = (j & 3) * 4;
        const uint32_t k = (j & 4) ? 2u : 0u;   // byte of the lane inside its word
        const uint32_t lo = perm8(st.s[w + 1], st.s[w], 0x0C0C0000u | ((4u + k) << 8) | k);
        const uint32_t hi = perm8(st.s[w + 3], st.s[w + 2], 0x00000C0Cu | ((4u + k) << 24) | (k << 16));
        d[j] = lo | hi;
    }
}
// natural -> paired lanes (clean)
SMI_HD void from_words(const uint32_t P[8], State &st) {
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t k = (uint32_t)w & 3u;
        if (SMI_CONST_P(P[(w >> 2) + 4]) && SMI_CONST_P(P[w >> 2]))
            st.s[w] = ((P[w >> 2] >> (8 * k)) & 0xFFu) | (((P[(w >> 2) + 4] >> (8 * k)) & 0xFFu) << 16);
        else
            st.s[w] = perm8(P[(w >> 2) + 4], P[w >> 2], 0x0C000C00u | ((4u + k) << 16) | k);
    }
}
SMI_HD uint32_t add_bytes(uint32_t a, uint32_t b) {   // four independent sums mod 256
    const uint32_t k7F = vreg(0x7F7F7F7Fu), k80 = vreg(0x80808080u);
    return ((a & k7F) + (b & k7F)) ^ ((a ^ b) & k80);
}
// (a ^ f) + m per byte: the XOR of the absorb recurrence folded into the two halves of the byte add
// (5 ops instead of 1 + 5): low seven bits (a ^ f) & 7F + m & 7F, top bits (a ^ f ^ m) & 80
SMI_HD uint32_t xor_add_bytes(uint32_t a, uint32_t f, uint32_t m) {
    const uint32_t k7F = vreg(0x7F7F7F7Fu), k80 = vreg(0x80808080u);
    const uint32_t t = bitop3<0x28>(a, f, k7F) + (m & k7F);      // 0x28: (a ^ b) & c
    return bitop3<0x78>(t, xor3(a, f, m), k80);                   // 0x78: a ^ (b & c)
}
SMI_HD uint32_t rotl3_bytes(uint32_t x) { return bfi32(vreg(0xF8F8F8F8u), x << 3, x >> 5); }

// src/hash.rs:15-20 for one full chunk.  With v_i the value byte i takes when it is processed,
//     v_i = rotl3(s_i + m_i)               (i < 7)
//     v_i = rotl3((s_i ^ v_{i-7}) + m_i)   (i >= 7: the XOR from byte i-7 arrived earlier)
// and afterwards bytes 0..6 receive the XORs of bytes 25..31.  Byte i depends on byte i-7 only, so
// the four bytes of a word are independent and word j needs words j-2 and j-1.
SMI_HD void absorb32_words(uint32_t P[8], const uint32_t M[8]) {
    uint32_t V[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j == 0) V[j] = rotl3_bytes(add_bytes(P[j], M[j]));
        else V[j] = rotl3_bytes(xor_add_bytes(P[j], j == 1 ? V[0] << 24 : funnel(V[j - 1], V[j - 2], 8), M[j]));   // ^ v_{4j-7} .. v_{4j-4}
    }
    V[0] ^= funnel(V[7], V[6], 8);   // bytes 0..3 ^= v_25..v_28
    V[1] ^= V[7] >> 8;               // bytes 4..6 ^= v_29..v_31
#pragma unroll
    for (int j = 0; j < 8; j++) P[j] = V[j];
}

// A short last chunk of nw message words (nw even, < 8; the messages on this path are whole u64s):
// bytes 0..4nw-1 go through the same recurrence, and the XORs they send seven places ahead land in
// the untouched bytes 4nw .. 4nw+6 (never past byte 31, so nothing wraps onto bytes 0..6).
SMI_HD void absorb_partial_words(uint32_t P[8], const uint32_t *M, int nw) {
    uint32_t V[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j >= nw) break;
        uint32_t x = P[j];
        if (j == 1) x ^= V[0] << 24;
        if (j >= 2) x ^= funnel(V[j - 1], V[j - 2], 8);
        V[j] = rotl3_bytes(add_bytes(x, M[j]));
        P[j] = V[j];
    }
    P[nw] ^= nw >= 2 ? funnel(V[nw - 1], V[nw - 2], 8) : (V[0] << 24);   // bytes 4nw..4nw+3 ^= v_{4nw-7}..v_{4nw-4}
    P[nw + 1] ^= V[nw - 1] >> 8;                                          // bytes 4nw+4..4nw+6 ^= v_{4nw-3}..v_{4nw-1}
}

// initial state in natural layout (src/hash.rs:10-12)
struct InitWords {
    uint32_t p[8];
};
constexpr InitWords make_init_words() {
    const uint8_t pr[16] = SMI_PRIMES;
    InitWords w{};
    for (int j = 0; j < 8; j++)
        for (int k = 0; k < 4; k++) w.p[j] |= (uint32_t)pr[(4 * j + k) % 16] << (8 * k);
    return w;
}

// absorb one full 32-byte chunk given as 8 LE words, then mix (src/hash.rs:14-23); the state
// must be fully applied (no pending round constants)
SMI_HD void absorb_chunk32(State &st, const uint32_t m[8]) {
    uint32_t P[8];
    to_words(st, P);
    absorb32_words(P, m);
    from_words(P, st);
    mix(st);
}

// One Fiat-Shamir round of Fri::commit on a transcript kept as the sponge state after its whole chunks
// (only 32-byte roots are ever absorbed, src/fri.rs:131): absorb the root, append it to the proof (tag 0
// + 32 bytes, src/stream.rs:39-42) and, unless this is the last round, draw the challenge
// (src/fiat_shamir.rs:19-25: the first 8 digest bytes, unreduced).  Single lane.
SMI_HD void fs_absorb_root(uint32_t *fs_words, const uint32_t m[8], uint8_t *proof_slot, uint64_t *alpha_out) {
    State st;
    for (int i = 0; i < 16; i++) st.s[i] = fs_words[i];
    absorb_chunk32(st, m);
    for (int i = 0; i < 16; i++) fs_words[i] = st.s[i];
    if (proof_slot) {
        proof_slot[0] = 0;
        for (int i = 0; i < 32; i++) proof_slot[1 + i] = (uint8_t)(m[i >> 2] >> (8 * (i & 3)));
    }
    if (alpha_out) {
        for (int k = 0; k < 8; k++) mix(st);
        uint32_t d[8];
        to_words(st, d);
        *alpha_out = (uint64_t)d[0] | ((uint64_t)d[1] << 32);
    }
}

// Hash::from_field_elements(&[v as u64]) (src/hash.rs:32-35 as used by src/fri.rs:118-121):
// 8 message bytes (LE u64 of a u32 residue: the upper four are zero), 1 + 8 mixes.
// the 8-byte chunk of a leaf in natural layout: it touches bytes 0..14 -- v_0..v_7 as in
// absorb32_words (m_4..m_7 = 0), then bytes 8..14 ^= v_1..v_7
SMI_HD void leaf_absorb_words(uint32_t v, uint32_t P[8]) {
    constexpr InitWords I = make_init_words();
#pragma unroll
    for (int j = 0; j < 8; j++) P[j] = I.p[j];
    const uint32_t V0 = rotl3_bytes(add_bytes(P[0], v));
    const uint32_t V1 = rotl3_bytes(P[1] ^ (V0 << 24));
    P[0] = V0;
    P[1] = V1;
    P[2] ^= funnel(V1, V0, 8);
    P[3] ^= V1 >> 8;
}
SMI_HD void leaf_hash(uint32_t v, uint32_t d[8]) {
    uint32_t P[8];
    leaf_absorb_words(v, P);
    State st;
    from_words(P, st);
    const MixK K = mix_consts();
    mix_t<false>(st, K);
#pragma unroll 1
    for (int k = 0; k < 8; k++) mix_t<true>(st, K);
    flush(st);
    to_words(st, d);
}

// Hash::combine (src/hash.rs:41-46): 64 bytes = two chunks, 2 + 8 mixes.  UNROLL: how many of the 8
// closing mixes the compiler may schedule together -- 1 where the kernel is bound by VALU issue (code
// size, registers), more where a single wave per SIMD waits on the ring add's dependent chain and
// the next mix's S-boxes can start under it (merkle_top_kernel).
template <int UNROLL = 1> SMI_HD void node_hash(const uint32_t l[8], const uint32_t r[8], uint32_t d[8]) {
    constexpr InitWords I = make_init_words();
    uint32_t P[8];
#pragma unroll
    for (int j = 0; j < 8; j++) P[j] = I.p[j];
    absorb32_words(P, l);
    State st;
    from_words(P, st);
    const MixK K = mix_consts();
    mix(st, K);
    to_words(st, P);
    absorb32_words(P, r);
    from_words(P, st);
    mix_t<false>(st, K);
#pragma unroll UNROLL
    for (int k = 0; k < 8; k++) mix_t<true>(st, K);
    flush(st);
    to_words(st, d);
}

// ---- two hashes per state.  Word w holds byte w of hash X in lane 0 and byte w of hash Y in lane 1
// (32 words for the pair: the same 16 registers per hash).  S-box and linear mix are unchanged
// per word; the ring add loses both of its lane crossings -- each lane now runs the reference's
// own recurrence new[i] = s[i] + s[i+1] + new[i-1] from byte 0 to byte 31 -- so the byte-sum tree and
// the fix-ups of mix_t disappear: 88 instead of 102 VALU instructions per mix and hash.  The Merkle
// kernels hash leaves and nodes in pairs; single hashes (transcripts, the odd node) use State.
struct State2 {
    uint32_t s[32];
};
struct Consts2 {
    uint32_t rc[32], rc502[32];
};
constexpr Consts2 make_consts2() {
    const uint8_t rc[32] = SMI_RC;
    Consts2 c{};
    for (int w = 0; w < 32; w++) {
        c.rc[w] = (uint32_t)rc[w] * 0x00010001u;
        c.rc502[w] = ((502u * rc[w]) & 0xFFFFu) * 0x00010001u;
    }
    return c;
}
SMI_HD void flush2(State2 &st) {
    constexpr Consts2 C = make_consts2();
#pragma unroll
    for (int w = 0; w < 32; w++) st.s[w] += C.rc[w];
}
template <bool PENDING> SMI_HD void mix2_t(State2 &st, const MixK &K) {
    constexpr Consts2 C = make_consts2();
    uint32_t *s = st.s;
    const uint32_t kFE = K.kFE, kFF = K.kFF, k63 = K.k63;
    // A VOP3P instruction takes ONE scalar/literal source: with the multiplier in a VGPR the 32 different round-constant
    // addends can be that one (SGPRs, s_mov on the scalar unit) instead of being re-materialised into VGPRs by
    // v_mov inside the mix loop (10 per mix2 before, r03).
    const uint32_t k502 = K.k502;
    uint32_t r[32];
#pragma unroll
    for (int w = 0; w < 32; w++) {
        const uint32_t t = pk_mad_u16(s[w], SMI_CONST_P(s[w]) ? 0x01F601F6u : k502, PENDING ? C.rc502[w] : 0u);
        r[w] = sbox_tail(t, kFE);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t t0 = r[4 * q], t1 = r[4 * q + 1], t2 = r[4 * q + 2], t3 = r[4 * q + 3];
        const uint32_t T = lin_sum(t0, t1, t2, t3, k63);
        s[4 * q] = lin_out(T, t2, kFF);
        s[4 * q + 1] = lin_out(T, t1, kFF);
        s[4 * q + 2] = lin_out(T, t3, kFF);
        s[4 * q + 3] = lin_out(T, t0, kFF);
    }
    // src/hash.rs:77-81 as written, both lanes at once; a lane never exceeds 32*2*255 + 1020 < 2^16.
    // One v_add3_u32 per word.  Measured r03 (profiles/r03_d_ringadd_ab.log): spelling it as two plain adds -- the pair
    // sums off the chain, one full-rate add per word on it (pair_sum) -- wins 7 % in a bare mix loop at 4 waves per SIMD
    // (tools/ubench_mix.hip) and LOSES 6 % in the Merkle kernels, which run 8 waves per SIMD at 62 VGPRs: there the
    // chain's latency is hidden already and the 32 extra instructions per mix are pure issue cost (prove 11.93 -> 12.66 ms).
    uint32_t N[32];
    N[0] = add3(s[0], s[1], s[31]);
#pragma unroll
    for (int w = 1; w < 31; w++) N[w] = add3(N[w - 1], s[w], s[w + 1]);
    N[31] = add3(s[31], N[0], N[30]);
#pragma unroll
    for (int w = 0; w < 32; w++) s[w] = N[w];
}
template <bool PENDING> SMI_HD void mix2_t(State2 &st) { mix2_t<PENDING>(st, mix_consts()); }
SMI_HD void mix2(State2 &st, const MixK &K) {
    mix2_t<false>(st, K);
    flush2(st);
}
SMI_HD void from_words2(const uint32_t X[8], const uint32_t Y[8], State2 &st) {
#pragma unroll
    for (int w = 0; w < 32; w++) {
        const uint32_t k = (uint32_t)w & 3u;
        if (SMI_CONST_P(X[w >> 2]) && SMI_CONST_P(Y[w >> 2]))   // untouched initial-state words of a leaf: fold
            st.s[w] = ((X[w >> 2] >> (8 * k)) & 0xFFu) | (((Y[w >> 2] >> (8 * k)) & 0xFFu) << 16);
        else
            st.s[w] = perm8(Y[w >> 2], X[w >> 2], 0x0C000C00u | ((4u + k) << 16) | k);
    }
}
SMI_HD void to_words2(const State2 &st, uint32_t X[8], uint32_t Y[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // two perms gather (X_4j, X_4j+1 | Y_4j, Y_4j+1) and (X_4j+2, X_4j+3 | Y_4j+2, Y_4j+3), two more split them
        // into the X word and the Y word: 4 ops per pair of words instead of 6
        const uint32_t *q = st.s + 4 * j;
        const uint32_t a = perm8(q[1], q[0], 0x06020400u), b = perm8(q[3], q[2], 0x06020400u);
        X[j] = perm8(b, a, 0x05040100u);
        Y[j] = perm8(b, a, 0x07060302u);
    }
}
// two leaves / two nodes at once: the same digests as leaf_hash / node_hash on each
SMI_HD void leaf_hash2(uint32_t v0, uint32_t v1, uint32_t d0[8], uint32_t d1[8]) {
    uint32_t X[8], Y[8];
    leaf_absorb_words(v0, X);
    leaf_absorb_words(v1, Y);
    State2 st;
    from_words2(X, Y, st);
    const MixK K = mix_consts();
    mix2_t<false>(st, K);
#pragma unroll 1
    for (int k = 0; k < 8; k++) mix2_t<true>(st, K);
    flush2(st);
    to_words2(st, d0, d1);
}
SMI_HD void node_hash2(const uint32_t l0[8], const uint32_t r0[8], const uint32_t l1[8], const uint32_t r1[8], uint32_t d0[8],
                       uint32_t d1[8]) {
    constexpr InitWords I = make_init_words();
    uint32_t X[8], Y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) X[j] = Y[j] = I.p[j];
    absorb32_words(X, l0);
    absorb32_words(Y, l1);
    State2 st;
    from_words2(X, Y, st);
    const MixK K = mix_consts();
    mix2(st, K);
    to_words2(st, X, Y);
    absorb32_words(X, r0);
    absorb32_words(Y, r1);
    from_words2(X, Y, st);
    mix2_t<false>(st, K);
#pragma unroll 1
    for (int k = 0; k < 8; k++) mix2_t<true>(st, K);
    flush2(st);
    to_words2(st, d0, d1);
}

// ---- row leaves: Hash::from_field_elements(&row) (src/hash.rs:32-35) for a trace row of W u32
// residues, i.e. the 8W-byte message of their LE u64s.  W = 4 is exactly one 32-byte chunk (1 + 8
// mixes, the cost of a single-element leaf); other widths take full chunks of four elements and a
// short last chunk.  Build-defined leaf rule: the reference itself only ever
// hashes one element per leaf (src/fri.rs:118-121).
SMI_HD void row_chunk_words(const uint32_t *v, int count, uint32_t P[8]) {   // absorb `count` (1..4) elements into P
    uint32_t M[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        M[2 * j] = j < count ? v[j] : 0u;
        M[2 * j + 1] = 0u;
    }
    if (count == 4) absorb32_words(P, M);
    else absorb_partial_words(P, M, 2 * count);
}
SMI_HD void row_hash(const uint32_t *v, int W, uint32_t d[8]) {
    constexpr InitWords I = make_init_words();
    uint32_t P[8];
#pragma unroll
    for (int j = 0; j < 8; j++) P[j] = I.p[j];
    State st;
    int done = 0;
    bool first = true;
    do {                                             // one pass per chunk (none for an empty row)
        const int count = W - done < 4 ? W - done : 4;
        if (count <= 0) break;
        if (!first) to_words(st, P);
        row_chunk_words(v + done, count, P);
        from_words(P, st);
        mix(st);
        first = false;
        done += count;
    } while (done < W);
    if (first) from_words(P, st);
    for (int k = 0; k < 8; k++) mix(st);
    to_words(st, d);
}
// two rows of W <= 4 elements at once (the Merkle kernels' fast path)
SMI_HD void row_hash2(const uint32_t *v0, const uint32_t *v1, int W, uint32_t d0[8], uint32_t d1[8]) {
    constexpr InitWords I = make_init_words();
    uint32_t X[8], Y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) X[j] = Y[j] = I.p[j];
    row_chunk_words(v0, W, X);
    row_chunk_words(v1, W, Y);
    State2 st;
    from_words2(X, Y, st);
    const MixK K = mix_consts();
    mix2_t<false>(st, K);
#pragma unroll 1
    for (int k = 0; k < 8; k++) mix2_t<true>(st, K);
    flush2(st);
    to_words2(st, d0, d1);
}

// Hash::from_bytes for an arbitrary message (src/hash.rs:7-30); single lane, used by the
// Fiat-Shamir and index-sampling kernels (transcripts of a few hundred bytes).
SMI_HD void hash_bytes(const uint8_t *msg, size_t len, uint32_t d[8]) {
    State st;
    init(st);
    for (size_t off = 0; off < len; off += 32) {
        const size_t clen = len - off < 32 ? len - off : 32;
        for (size_t i = 0; i < clen; i++) {
            const int pos = (int)i;
            uint32_t v = (get_byte(st, pos) + msg[off + i]) & 0xFFu;
            v = ((v << 3) | (v >> 5)) & 0xFFu;
            set_byte(st, pos, v);
            xor_byte(st, (pos + 7) & 31, v);
        }
        mix(st);
    }
    for (int k = 0; k < 8; k++) mix(st);
    to_words(st, d);
}

}  // namespace hashc
