// hash_hex.h -- Hash::combine (src/hash.rs:41-46) with one hash spread over a ROW of sixteen lanes (device only).
//
// The narrowest levels of a Merkle tree (at most 64 nodes in a 1024-lane workgroup) cost one node-hash LATENCY each:
// what counts there is the number of dependent instructions one lane runs, not the lane-instructions spent.  The quad
// form (hash_quad.h) runs ~850 per node; here lane w of an aligned row of sixteen holds ONE word of the paired-lane
// state (hash_core.h: state byte w in bits 0..7, state byte 16+w in bits 16..23), and
//   * the S-box (src/hash.rs:88-94) is the same three instructions, on one word;
//   * the 4-byte linear mix (src/hash.rs:64-75) works on bytes 4g..4g+3, i.e. on the four lanes of a quad: two
//     quad_perm XOR steps for the group sum, one more for the partner byte;
//   * the sequential ring add (src/hash.rs:77-81) is a prefix sum: s'[i] = s[31] + sum_{j<=i} (s[j] + s[j+1]) for
//     i < 31, s'[31] = s[31] + s'[0] + s'[30].  Both 16-bit lanes scan at once over the row (four row_shr adds);
//     the low lane's total s'[15] is what the high lane starts from, lane 0 brings s[31] in and lane 15 closes the
//     ring (see mix);
//   * the absorb recurrence (src/hash.rs:15-20) sends byte i's value seven places ahead: five dependent stages of
//     seven positions each (0..6, 7..13, ... 28..31), a row rotation by seven carrying the values between them.
// ~25 instructions per mix and ~60 per absorbed chunk instead of 58 and ~85+: ~400 per node hash.
// Same digests as hashc::node_hash: on the CPU the emulator runs this very code with `word` = sixteen lanes stepped in
// lockstep (SMI_HEX_EMU, csrc/emu.cpp), on the device the Merkle tests of the GPU suite go through it.
#pragma once
#include "hash_core.h"

#if defined(__HIPCC__) || defined(SMI_HEX_EMU)
namespace hashx {

#if !defined(SMI_HEX_EMU)
#define SMI_XD __device__ __forceinline__
typedef uint32_t word;   // one lane's 32-bit value
// lane i of every aligned row of 16 reads lane (i - N) mod 16 of the same row (DPP row_ror:N).  bound_ctrl is set on every
// move although no lane ever reads out of range: without it the compiler has to materialise the `old` operand (a v_mov per
// move) and cannot fold the move into the DPP operand of the instruction that consumes it
template <int N> SMI_XD word rot(word x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x120 | N, 0xF, 0xF, true);
#else
    return x;   // host pass of the compiler only: never executed
#endif
}
// lane i reads lane i - N, lanes below N read zero (DPP row_shr:N, bound_ctrl)
template <int N> SMI_XD word shr(word x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 | N, 0xF, 0xF, true);
#else
    return x;
#endif
}
template <int P0, int P1, int P2, int P3> SMI_XD word quad(word x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
#else
    return x;
#endif
}
SMI_XD word lane_in_row(uint32_t lane_id) { return lane_id & 15u; }
SMI_XD word mask_range(word w, uint32_t lo, uint32_t hi) { return w >= lo && w < hi ? ~0u : 0u; }   // all-ones where lo <= w < hi
SMI_XD word table16(word w, const uint32_t (&t)[16]) {
    word r = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) r = w == k ? t[k] : r;
    return r;
}
SMI_XD word perm8v(word hi, word lo, word sel) { return hashc::perm8(hi, lo, sel); }
SMI_XD word lshl_or(word a, int sh, word b) { return (a << sh) | b; }
SMI_XD word lshl_add(word a, int sh, word b) { return (a << sh) + b; }
#endif

// per-lane constants, set up once per kernel from w = lane & 15
struct Lane {
    word w;
    word init;           // the lane's word of the initial state (PRIMES)
    word rc, rc502;      // round-constant pair of word w, and the S-box addend that applies it when pending
    word is0, is15, lt7; // all-ones where w == 0, w == 15, w < 7
    word st[5];          // absorb stage j: the 16-bit lanes of this word that hold positions 7j .. 7j+6
    word wrap;           // positions 25 .. 31
    word sel;            // byte selector: natural words (w>>2, 4+(w>>2)) -> the lane's message / state word
};
SMI_XD Lane make_lane(uint32_t lane_id) {
    constexpr hashc::Consts C = hashc::make_consts();
    const uint32_t pr[16] = SMI_PRIMES;
    uint32_t init[16];
    for (int k = 0; k < 16; k++) init[k] = pr[k] * 0x00010001u;
    Lane L;
    L.w = lane_in_row(lane_id);
    L.init = table16(L.w, init);
    L.rc = table16(L.w, C.rc);
    L.rc502 = table16(L.w, C.rc502);
    L.is0 = mask_range(L.w, 0, 1);
    L.is15 = mask_range(L.w, 15, 16);
    L.lt7 = mask_range(L.w, 0, 7);
    // position p = w (low lane) and 16 + w (high lane)
    L.st[0] = mask_range(L.w, 0, 7) & word(0x0000FFFFu);
    L.st[1] = mask_range(L.w, 7, 14) & word(0x0000FFFFu);
    L.st[2] = (mask_range(L.w, 14, 16) & word(0x0000FFFFu)) | (mask_range(L.w, 0, 5) & word(0xFFFF0000u));
    L.st[3] = mask_range(L.w, 5, 12) & word(0xFFFF0000u);
    L.st[4] = mask_range(L.w, 12, 16) & word(0xFFFF0000u);
    L.wrap = mask_range(L.w, 9, 16) & word(0xFFFF0000u);
    L.sel = word(0x0C040C00u) + (L.w & word(3u)) * word(0x00010001u);   // bytes k of lo and of hi -> bits 0..7 and 16..23
    return L;
}

// every position's value moved seven positions ahead (mod 32): a row rotation by seven, and the lanes it wraps
// into (w < 7) take the other 16-bit lane
SMI_XD word ahead7(word v, const Lane &L) {
    const word z = rot<7>(v);
    return hashc::bfi32(L.lt7, hashc::funnel16(z, z), z);
}

// src/hash.rs:15-20 for one 32-byte chunk: m = the lane's two message bytes (positions w and 16 + w, clean lanes);
// x = the lane's state word, fully applied (no pending round constants); returns it with clean lanes
SMI_XD word absorb(word x, word m, const Lane &L) {
    const word kFF = word(0x00FF00FFu);
    x = x & kFF;
    word inc = word(0u);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const word t = ((x ^ inc) + m) & kFF;
        const word r = lshl_or(t, 3, t >> 5) & kFF;                // rotl 3 of both bytes
        x = hashc::bfi32(L.st[j], r, x);
        if (j < 4) inc = ahead7(x & L.st[j], L);                    // lands on the positions of stage j + 1 (and, from stage 3, on 0..2: not taken)
    }
    return x ^ ahead7(x & L.wrap, L);                               // bytes 0..6 ^= v_25 .. v_31
}

// src/hash.rs:59-86 on the lane's word; PENDING as in hashc::mix_t (the round constants of the previous mix folded into
// this S-box, its own left pending)
template <bool PENDING> SMI_XD word mix(word x, const Lane &L) {
    using namespace hashc;
    const word kFE = word(0x00FE00FEu), kFF = word(0x00FF00FFu), k63 = word(0x00630063u);
    const word t = pk_mad_u16(x, word(0x01F601F6u), PENDING ? L.rc502 : word(0u));
    const word r = bfi32(kFE, t, t >> 8);
    // linear mix: lane k of a quad takes T ^ r[{2, 1, 3, 0}[k]], T = the quad's XOR (^0x63 deferred to here)
    const word a = r ^ quad<1, 0, 3, 2>(r);
    const word T = a ^ quad<2, 3, 0, 1>(a);
    const word s = xor3(T, quad<2, 1, 3, 0>(r), k63) & kFF;
    // ring add.  With a_w / b_w the low / high byte of lane w:  d_w = s_w + s_{w+1} on both lanes for w <= 14;
    // lane 15: low a_15 + b_0 (s[15] + s[16]), high b_15 + new[0] with new[0] = a_0 + a_1 + b_15 (the ring closes);
    // lane 0 also brings in s[31] = b_15 on the low lane.  An inclusive scan over the row then gives new[w] on the low
    // lanes and new[16 + w] - new[15] on the high ones; new[15] is the row total of the low lanes.
    const word nx = rot<15>(s);
    const word d = s + nx;
    const word sh = s >> 16;
    const word tt = sh + rot<15>(d);                                // lane 15: b_15 + a_0 + a_1
    const word d15 = s + lshl_or(tt, 16, nx >> 16);                // low + b_0, high + new[0]
    word D = bfi32(L.is15, d15, d) + (rot<1>(sh) & L.is0);
    word tot = D + rot<8>(D);
    D = D + shr<1>(D);
    tot = tot + rot<4>(tot);
    D = D + shr<2>(D);
    tot = tot + quad<2, 3, 0, 1>(tot);
    D = D + shr<4>(D);
    tot = tot + quad<1, 0, 3, 2>(tot);
    D = D + shr<8>(D);
    return lshl_add(tot, 16, D);
}

// Hash::combine over a row of sixteen: ml / mr = the lane's two bytes (positions w and 16 + w) of the left / right
// child digest, clean lanes.  Returns the lane's digest bytes w (bits 0..7) and 16 + w (bits 16..23); the other bits
// are not meaningful.  All sixteen lanes of the row must be active.
SMI_XD word node_hash(word ml, word mr, const Lane &L) {
    word x = absorb(L.init, ml, L);
    x = mix<false>(x, L) + L.rc;
    x = absorb(x, mr, L);
    x = mix<false>(x, L);
#pragma unroll
    for (int k = 0; k < 8; k++) x = mix<true>(x, L);
    return x + L.rc;
}
// the lane's message word from the natural-layout words j = w >> 2 and 4 + j of a digest
SMI_XD word message(word lo, word hi, const Lane &L) { return perm8v(hi, lo, L.sel); }

// One Fiat-Shamir round of Fri::commit over the row (hashc::fs_absorb_root lane by lane: 3.2 us of a tree's last launch when
// a single lane runs it, profiles/r03_z_tophash_ubench.log).  x = the lane's word of the transcript's sponge state (fully
// applied, as hashc::fs_absorb_root keeps it in memory), m = its two bytes of the root: absorb and mix (src/hash.rs:14-23).
SMI_XD word fs_absorb(word x, word m, const Lane &L) { return mix<false>(absorb(x, m, L), L) + L.rc; }
// FiatShamir::challenge (src/fiat_shamir.rs:19-25) of the state x: the eight closing mixes on a copy; bits 0..7 of lanes
// 0..7 are the challenge's bytes, little end first
SMI_XD word fs_challenge(word x, const Lane &L) {
    x = mix<false>(x, L);
#pragma unroll
    for (int k = 0; k < 7; k++) x = mix<true>(x, L);
    return x + L.rc;
}
#if !defined(SMI_HEX_EMU) && defined(__HIPCC__)
// the low bytes of lanes 0..7 of the wave's first row as one u64 (wave-uniform)
__device__ __forceinline__ uint64_t low_bytes_u64(uint32_t y) {
    uint64_t a = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) a |= (uint64_t)((uint32_t)__builtin_amdgcn_readlane((int)y, k) & 0xFFu) << (8 * k);
    return a;
}
// hashc::fs_absorb_root by lanes 0..15 of a wave (all sixteen active, no other row of the wave inside this call): absorb the
// root, append it to the proof (tag 0 + 32 bytes), draw the challenge unless alpha_out == nullptr
__device__ __forceinline__ void fs_absorb_root(uint32_t *fs_words, uint32_t m, const Lane &L, uint8_t *proof_slot, uint64_t *alpha_out) {
    const uint32_t x = fs_absorb(fs_words[L.w], m, L);
    fs_words[L.w] = x;
    if (proof_slot) {
        if (L.w == 0) proof_slot[0] = 0;
        proof_slot[1 + L.w] = (uint8_t)m;
        proof_slot[17 + L.w] = (uint8_t)(m >> 16);
    }
    if (alpha_out) {
        const uint64_t a = low_bytes_u64(fs_challenge(x, L));
        if (L.w == 0) *alpha_out = a;
    }
}
#endif

}  // namespace hashx
#endif
