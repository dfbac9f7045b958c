// hash_core.h -- the reference's 32-byte hash (src/hash.rs:7-99) for one lane (host+device).
//
// State layout ("paired lanes"): 16 u32 words, word w holds state byte w in bits 0..7 and
// state byte w+16 in bits 16..23; bits 8..15 / 24..31 are headroom.  With it
//   * the byte S-box  rotl1(b*251)^0x63  is one 24-bit multiply by 502 on two bytes at once
//     (bit 8 of 502*b is the rotated-out bit), src/hash.rs:88-94;
//   * the 4-byte linear mix (src/hash.rs:64-75) is plain XORs of whole words: groups g and
//     g+4 share words 4g..4g+3;
//   * the sequential in-place ring add (src/hash.rs:77-81), which is a prefix sum mod 256:
//        new[i] = P[i] + P[i+1] + old[31] - old[0]   (i <= 30),  P[i] = sum_{j<=i} old[j]
//        new[31] = old[31] + new[0] + new[30]
//     becomes a 15-add running sum over the words (lane sums stay < 2^16, so the two bytes
//     of a word never interfere), with the round constants (src/hash.rs:83-85,96-99)
//     folded into the per-word constants J_w.
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
    uint32_t J[16];   // J_0 = 0, J_w + J_{w+1} = RC pair of word w (per lane, mod 256)
    uint32_t lo15;    // RC[15]
    uint32_t hi15;    // RC[31] - RC[0] - RC[30]  (mod 256, kept positive)
};
constexpr Consts make_consts() {
    const uint8_t rc[32] = SMI_RC;
    Consts c{};
    uint32_t jl = 0, jh = 0;  // J_w lanes
    for (int w = 0; w < 16; w++) {
        c.J[w] = jl | (jh << 16);
        jl = (uint32_t)(rc[w] - jl) & 0xFFu;
        jh = (uint32_t)(rc[w + 16] - jh) & 0xFFu;
    }
    c.lo15 = rc[15];
    c.hi15 = (uint32_t)(rc[31] + 512 - rc[0] - rc[30]) & 0xFFu;
    return c;
}

struct State {
    uint32_t s[16];
};

SMI_HD void init(State &st) {
    const uint8_t pr[16] = SMI_PRIMES;
#pragma unroll
    for (int w = 0; w < 16; w++) st.s[w] = (uint32_t)pr[w] * 0x00010001u;  // bytes w and w+16 are both PRIMES[w]
}

// src/hash.rs:59-86 on clean lanes (each lane <= 255); leaves clean lanes.
SMI_HD void mix(State &st) {
    constexpr Consts C = make_consts();
    uint32_t *s = st.s;
    // (1) S-box: t = 502*b; rotl1(251*b mod 256) = (t & 0xFE) | bit 8 of t.  XOR 0x63 is deferred
    // through the linear layer (it XORs three bytes, so the constant passes through unchanged).
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t t = SMI_MUL24(s[w], 502u);
        s[w] = (t & 0x00FE00FEu) | ((t >> 8) & 0x00010001u);
    }
    // (2) linear mix: new = (t0^t1^t2^t3) ^ {t2, t1, t3, t0}, then the deferred ^0x63
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t0 = s[4 * q], t1 = s[4 * q + 1], t2 = s[4 * q + 2], t3 = s[4 * q + 3];
        const uint32_t T = t0 ^ t1 ^ t2 ^ t3 ^ 0x00630063u;
        s[4 * q] = T ^ t2;
        s[4 * q + 1] = T ^ t1;
        s[4 * q + 2] = T ^ t3;
        s[4 * q + 3] = T ^ t0;
    }
    // (3) ring add as a prefix sum + (4) round constants
    const uint32_t old0 = s[0] & 0xFFFFu, old16 = s[0] >> 16, old31 = s[15] >> 16;
    const uint32_t c = old31 + 256u - old0;  // old[31] - old[0] mod 256, positive
    const uint32_t cc = c * 0x00010001u;
    uint32_t S[16];
    S[0] = s[0];
#pragma unroll
    for (int w = 1; w < 16; w++) S[w] = S[w - 1] + s[w];       // lane0: P[w]; lane1: P[16+w] - P[15]
    const uint32_t p15 = S[15] & 0xFFFFu;
    const uint32_t p15s = p15 << 16;
    uint32_t G[16];
#pragma unroll
    for (int w = 0; w < 16; w++) G[w] = S[w] + p15s + C.J[w];  // lanes: P[w]+J, P[16+w]+J
#pragma unroll
    for (int w = 0; w < 15; w++) s[w] = G[w] + G[w + 1] + cc;  // new[w], new[16+w] incl. round constants
    const uint32_t lo = 2u * p15 + old16 + c + C.lo15;          // new[15] = P[15] + P[16] + c
    const uint32_t hi = old31 + (s[0] & 0xFFFFu) + (s[14] >> 16) + C.hi15;  // new[31]
    s[15] = (lo & 0xFFu) | (hi << 16);
#pragma unroll
    for (int w = 0; w < 16; w++) s[w] &= 0x00FF00FFu;
}

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

// digest as 8 little-endian u32 words (bytes 4j..4j+3 in d[j])
SMI_HD void to_words(const State &st, uint32_t d[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int w = (j & 3) * 4, sh = (j & 4) ? 16 : 0;
        d[j] = ((st.s[w] >> sh) & 0xFFu) | (((st.s[w + 1] >> sh) & 0xFFu) << 8) | (((st.s[w + 2] >> sh) & 0xFFu) << 16) |
               (((st.s[w + 3] >> sh) & 0xFFu) << 24);
    }
}

// absorb one full 32-byte chunk given as 8 LE words, then mix (src/hash.rs:14-23)
SMI_HD void absorb_chunk32(State &st, const uint32_t m[8]) {
#pragma unroll
    for (int i = 0; i < 32; i++) absorb_byte(st, i, (m[i >> 2] >> (8 * (i & 3))) & 0xFFu);
    mix(st);
}

// Hash::from_field_elements(&[v as u64]) (src/hash.rs:32-35 as used by src/fri.rs:118-121):
// 8 message bytes (LE u64 of a u32 residue: the upper four are zero), 1 + 8 mixes.
SMI_HD void leaf_hash(uint32_t v, uint32_t d[8]) {
    State st;
    init(st);
#pragma unroll
    for (int i = 0; i < 8; i++) absorb_byte(st, i, i < 4 ? ((v >> (8 * i)) & 0xFFu) : 0u);
    mix(st);
#pragma unroll 1
    for (int k = 0; k < 8; k++) mix(st);
    to_words(st, d);
}

// Hash::combine (src/hash.rs:41-46): 64 bytes = two chunks, 2 + 8 mixes.
SMI_HD void node_hash(const uint32_t l[8], const uint32_t r[8], uint32_t d[8]) {
    State st;
    init(st);
    absorb_chunk32(st, l);
    absorb_chunk32(st, r);
#pragma unroll 1
    for (int k = 0; k < 8; k++) mix(st);
    to_words(st, d);
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
