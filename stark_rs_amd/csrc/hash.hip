// hash.hip -- leaf-parallel hash and Merkle tree kernels (gfx950).
//
// Replaces Hash::from_field_elements / Hash::combine (reference src/hash.rs:32-46) as they
// are used per codeword element and per tree node by Fri::commit (src/fri.rs:118-127) and
// MerkleTree::new (src/merkle.rs:11-38).  Hashes run in VGPRs, two per lane at a time (State2 of
// hash_core.h; a single one where a level is odd).  The tree is kept whole on the device
// (`nodes` of src/merkle.rs:6, levels back to back) so `open` is a gather and nothing is rebuilt.
//
// merkle_sub_kernel: every lane owns 2^K adjacent inputs and builds their K-level subtree
// on its own (2^K leaf hashes + 2^K - 1 node hashes at full lane occupancy, no cross-lane
// traffic); child digests wait in an LDS stash laid out [slot][word][thread] (bank =
// thread, conflict free).  merkle_top_kernel: one workgroup finishes a tree from its last
// <= 2048 digests.  Roofline: integer VALU issue (~0.9k instructions per leaf hash, ~1.2k per node
// hash), not HBM -- DESIGN.md section 3.
#include "fri_core.h"
#include "hash_core.h"
#include "hash_quad.h"
#include "hash_hex.h"
#include "internal.h"

#ifndef SMI_HASH_THREADS
#define SMI_HASH_THREADS 256   // tuning builds: -DSMI_HASH_THREADS=128 | 512
#endif
#define SMI_ROW_MAX 64   // columns per row leaf

__device__ __forceinline__ size_t level_offset(size_t n, uint32_t lvl) { return 2 * n - ((2 * n) >> lvl); }

// KT > 0: the number of levels per lane is a compile-time constant (the loops below unroll, the stash
// slots become constants); KT = 0: taken from the argument.  ROWS: the row-leaf path (runtime width) is
// compiled in -- the element-leaf instantiation the prover lives in carries neither.
// The four leaves of a lane when they are computed rather than read (LeafSrc, internal.h): stored to the codeword and returned.
template <int LEAF> __device__ __forceinline__ void leaf_values(const LeafSrc &src, const uint32_t *elems, size_t first, uint32_t (&v)[4]) {
    if constexpr (LEAF == LEAF_LOAD) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = elems[first + k];
    } else if constexpr (LEAF == LEAF_FOLD) {
        const uint32_t ah_m = fold_alpha_half(*src.alpha, src.inv2_m, src.F);
        const uint4 lo = *(const uint4 *)(src.lo + first), hi = *(const uint4 *)(src.hi + first);     // first is a multiple of 4
        v[0] = fold_element(lo.x, hi.x, (uint32_t)first, ah_m, src.inv2_m, src.S, src.F);
        v[1] = fold_element(lo.y, hi.y, (uint32_t)first + 1, ah_m, src.inv2_m, src.S, src.F);
        v[2] = fold_element(lo.z, hi.z, (uint32_t)first + 2, ah_m, src.inv2_m, src.S, src.F);
        v[3] = fold_element(lo.w, hi.w, (uint32_t)first + 3, ah_m, src.inv2_m, src.S, src.F);
        *(uint4 *)(src.cw_out + first) = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
        v[0] = v[1] = v[2] = v[3] = 0;
#pragma unroll
        for (uint32_t c = 0; c < SMI_LEAF_COMBINE_MAX; c++) {
            if (c >= src.n_cols) break;
            const uint32_t w_m = src.weights_m[c];   // (weights[c] mod p) in Montgomery form, wave-uniform
            const uint4 x = *(const uint4 *)(src.cols + c * src.stride + first);
            v[0] = fp_add(v[0], mont_mul(x.x, w_m, src.F), src.F.p);
            v[1] = fp_add(v[1], mont_mul(x.y, w_m, src.F), src.F.p);
            v[2] = fp_add(v[2], mont_mul(x.z, w_m, src.F), src.F.p);
            v[3] = fp_add(v[3], mont_mul(x.w, w_m, src.F), src.F.p);
        }
        *(uint4 *)(src.cw_out + first) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

template <bool FROM_ELEMS, int KT, bool ROWS, int LEAF = LEAF_LOAD>
__global__ __launch_bounds__(SMI_HASH_THREADS) void merkle_sub_kernel(const uint32_t *__restrict__ elems, uint4 *nodes,
                                                                        size_t n, uint32_t lvl_in, size_t count_in,
                                                                        uint32_t K_arg, size_t elem_stride, size_t node_stride,
                                                                        uint32_t row_cols, size_t row_stride, const LeafSrc src) {
    static_assert(LEAF == LEAF_LOAD || (FROM_ELEMS && KT == 2 && !ROWS), "computed leaves: the four-leaves-per-lane kernel only");
    const uint32_t K = KT ? (uint32_t)KT : K_arg;
    extern __shared__ __attribute__((aligned(16))) uint32_t stash[];  // [1<<K][8][SMI_HASH_THREADS]
    // blockIdx.y = tree of a batch of equally sized trees (e.g. the columns of a trace)
    elems += (size_t)blockIdx.y * elem_stride;
    nodes += (size_t)blockIdx.y * node_stride;
    const uint32_t tid = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * SMI_HASH_THREADS + tid;
    const size_t n_threads = count_in >> K;
    if (t >= n_threads) return;  // no barriers below: lanes are independent
    const uint32_t per = 1u << K;
    const size_t first = t << K;
    // Hashes go two at a time (hash_core.h, State2); K >= 1 makes every level but the last even.
    auto put = [&](uint4 *dst, uint32_t slot, const uint32_t (&d)[8]) {
        dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
        dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
#pragma unroll
        for (int w = 0; w < 8; w++) stash[(slot * 8 + w) * SMI_HASH_THREADS + tid] = d[w];
    };
    auto get = [&](uint32_t slot, uint32_t (&d)[8]) {
#pragma unroll
        for (int w = 0; w < 8; w++) d[w] = stash[(slot * 8 + w) * SMI_HASH_THREADS + tid];
    };
    if constexpr (KT == 2 && !ROWS) {
        // The shape the prover lives in, written out: four inputs per lane, two levels.  Only the first
        // pair of leaf digests waits in the stash (while the second pair is hashed); everything else goes
        // from the registers that produced it into the hash that consumes it.
        auto store = [&](uint4 *dst, const uint32_t (&d)[8]) {
            dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
            dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
        };
        uint32_t l0[8], r0[8], l1[8], r1[8];
        if constexpr (FROM_ELEMS) {
            uint32_t ev[4];
            leaf_values<LEAF>(src, elems, first, ev);
            {
                uint32_t d0[8], d1[8];
                hashc::leaf_hash2(ev[0], ev[1], d0, d1);
                put(nodes + 2 * first, 0, d0);
                put(nodes + 2 * (first + 1), 1, d1);
            }
            hashc::leaf_hash2(ev[2], ev[3], l1, r1);
            store(nodes + 2 * (first + 2), l1);
            store(nodes + 2 * (first + 3), r1);
            get(0, l0);
            get(1, r0);
        } else {
            const uint4 *src = nodes + 2 * (level_offset(n, lvl_in) + first);
            auto load = [&](int i, uint32_t (&d)[8]) {
                const uint4 a = src[2 * i], b = src[2 * i + 1];
                d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
            };
            load(0, l0); load(1, r0); load(2, l1); load(3, r1);
        }
        uint32_t n0[8], n1[8], top[8];
        hashc::node_hash2(l0, r0, l1, r1, n0, n1);
        uint4 *dst1 = nodes + 2 * (level_offset(n, lvl_in + 1) + (t << 1));
        store(dst1, n0);
        store(dst1 + 2, n1);
        hashc::node_hash(n0, n1, top);
        store(nodes + 2 * (level_offset(n, lvl_in + 2) + t), top);
        return;
    }
    if (FROM_ELEMS && ROWS && row_cols) {
        // row leaves: leaf i = Hash::from_field_elements(row i) over row_cols <= 4 columns row_stride
        // apart (wider rows are hashed by row_hash_kernel and enter as digests)
        for (uint32_t i = 0; i < per; i += 2) {
            uint32_t d0[8], d1[8];
            if (i + 1 < per) {
                uint32_t r0[4], r1[4];
                for (uint32_t c = 0; c < 4; c++) {
                    r0[c] = c < row_cols ? elems[c * row_stride + first + i] : 0u;
                    r1[c] = c < row_cols ? elems[c * row_stride + first + i + 1] : 0u;
                }
                hashc::row_hash2(r0, r1, (int)row_cols, d0, d1);
                put(nodes + 2 * (first + i), i, d0);
                put(nodes + 2 * (first + i + 1), i + 1, d1);
            } else {   // odd leaf of a level (K = 0 never happens here; kept for symmetry)
                uint32_t r0[4];
                for (uint32_t c = 0; c < 4; c++) r0[c] = c < row_cols ? elems[c * row_stride + first + i] : 0u;
                hashc::row_hash(r0, (int)row_cols, d0);
                put(nodes + 2 * (first + i), i, d0);
            }
        }
    } else if (FROM_ELEMS) {
        for (uint32_t i = 0; i < per; i += 2) {
            uint32_t d0[8], d1[8];
            if (i + 1 < per) {
                hashc::leaf_hash2(elems[first + i], elems[first + i + 1], d0, d1);
                put(nodes + 2 * (first + i), i, d0);
                put(nodes + 2 * (first + i + 1), i + 1, d1);
            } else {
                hashc::leaf_hash(elems[first + i], d0);
                put(nodes + 2 * (first + i), i, d0);
            }
        }
    } else {
        for (uint32_t i = 0; i < per; i++) {
            const uint4 *src = nodes + 2 * (level_offset(n, lvl_in) + first + i);
            const uint4 a = src[0], b = src[1];
#pragma unroll
            for (int w = 0; w < 8; w++) stash[(i * 8 + w) * SMI_HASH_THREADS + tid] = w < 4 ? (&a.x)[w] : (&b.x)[w - 4];
        }
    }
    for (uint32_t j = 1; j <= K; j++) {   // K and per are run-time values down here (KT == 0): nothing to unroll
        const uint32_t cnt = per >> j;
        uint4 *dst = nodes + 2 * (level_offset(n, lvl_in + j) + (t << (K - j)));
        for (uint32_t q = 0; q < cnt; q += 2) {
            uint32_t l0[8], r0[8], d0[8];
            get(2 * q, l0);
            get(2 * q + 1, r0);
            if (q + 1 < cnt) {
                uint32_t l1[8], r1[8], d1[8];
                get(2 * q + 2, l1);
                get(2 * q + 3, r1);
                hashc::node_hash2(l0, r0, l1, r1, d0, d1);
                put(dst + 2 * q, q, d0);        // slots q, q+1 < 2q: already consumed
                put(dst + 2 * q + 2, q + 1, d1);
            } else {
                hashc::node_hash(l0, r0, d0);
                put(dst + 2 * q, q, d0);
            }
        }
    }
}

// Upper levels: one workgroup takes a chunk of at most SMI_TOP_MAX adjacent digests of one level (or
// that many codeword elements) and builds every level above them down to the chunk's single root,
// a barrier per level, the current level held in LDS as [word][slot].  grid.x = chunk, grid.y =
// tree.  These levels are pure latency (one node hash deep each): as two-level launches of
// merkle_sub_kernel they cost ~6.5 us per level, here one hash latency (~3.8 us).
#ifndef SMI_TOP_UNROLL
#define SMI_TOP_UNROLL 8   // closing mixes of a node hash scheduled together (see hashc::node_hash)
#endif
#ifndef SMI_TOP_QUAD
#define SMI_TOP_QUAD 1     // levels with at most SMI_TOP_THREADS / 4 nodes hash over quads of lanes
#endif
#ifndef SMI_TOP_HEX
#define SMI_TOP_HEX 16     // levels with at most this many nodes hash over rows of sixteen lanes (hash_hex.h); 0: never.
#endif                     // 16 nodes = 256 lanes = one wave per SIMD: a level of a 1024-lane workgroup takes 1.0 us that way and
                           // 1.45 over quads; with 32 nodes (two waves per SIMD) both take 1.5, with 64 the rows lose, 2.5 to 1.5
                           // (tools/quad_hash_test.hip, profiles/r03_y_tophash_ubench.log)
#define SMI_TOP_MAX 2048
#define SMI_TOP_THREADS (SMI_TOP_MAX / 2)
// Optional epilogue of the launch that produces a tree's root: the Fiat-Shamir round of Fri::commit
// (hashc::fs_absorb_root) by lane 0 of the workgroup holding the root -- one launch fewer per FRI round.
struct TopHook {
    uint32_t *fs_words;    // nullptr: no epilogue
    uint8_t *proof_slot;
    uint64_t *alpha_out;
};
// One workgroup: leaf digests (FROM_ELEMS) or the digests of level lvl_in for positions [first, first + chunk),
// then every level above them down to the chunk's root; all of it written to `nodes`.  Returns the
// slot of each word row of buf ([8][SMI_TOP_MAX]) that holds the root.
// FOLD (with FROM_ELEMS): element first + i is not read but computed -- Fri::fold_codeword of the round before (LeafSrc,
// LEAF_FOLD: fri_core.h fold_element, what the stand-alone fri_fold_kernel runs) -- and stored to src.cw_out on the way.
template <bool FROM_ELEMS, bool FOLD = false>
__device__ __forceinline__ uint32_t top_chunk(const uint32_t *__restrict__ elems, uint4 *nodes, size_t n, uint32_t lvl_in, uint32_t chunk,
                                              size_t first, uint32_t row_cols, size_t row_stride, uint32_t *buf, const LeafSrc *src = nullptr) {
    const uint32_t tid = threadIdx.x;
    uint32_t d[8];
    uint32_t ah_m = 0;
    if constexpr (FOLD) ah_m = fold_alpha_half(*src->alpha, src->inv2_m, src->F);
    for (uint32_t i = tid; i < chunk; i += SMI_TOP_THREADS) {
        if (FROM_ELEMS) {
            if constexpr (FOLD) {
                const uint32_t v = fold_element(src->lo[first + i], src->hi[first + i], (uint32_t)(first + i), ah_m, src->inv2_m, src->S, src->F);
                src->cw_out[first + i] = v;
                hashc::leaf_hash(v, d);
            } else if (row_cols) {
                uint32_t row[4];
                for (uint32_t c = 0; c < 4; c++) row[c] = c < row_cols ? elems[c * row_stride + first + i] : 0u;
                hashc::row_hash(row, (int)row_cols, d);
            } else {
                hashc::leaf_hash(elems[first + i], d);
            }
            nodes[2 * (first + i)] = make_uint4(d[0], d[1], d[2], d[3]);
            nodes[2 * (first + i) + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        } else {
            const uint4 *src = nodes + 2 * (level_offset(n, lvl_in) + first + i);
            const uint4 a = src[0], b = src[1];
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        }
#pragma unroll
        for (int w = 0; w < 8; w++) buf[w * SMI_TOP_MAX + i] = d[w];
    }
    __syncthreads();
    uint32_t lvl = lvl_in, base = 0;
    const hashq::Lane lane = hashq::make_lane(tid);
    const hashx::Lane row = hashx::make_lane(tid);
    for (uint32_t cnt = chunk; cnt > 1; cnt >>= 1) {   // cnt, lvl are workgroup-uniform
        const uint32_t half = cnt >> 1;
        lvl++;
        if (SMI_TOP_HEX && half <= SMI_TOP_HEX && 16 * half <= SMI_TOP_THREADS) {
            // very few nodes: one hash per row of sixteen lanes, one state word per lane (hash_hex.h): 0.91 us per node hash
            // in a wave of its own where the quad form takes 1.42.  Lane w reads bytes w and 16 + w of each child (natural words
            // w >> 2 and 4 + (w >> 2)) and writes those two bytes of the digest.  Slots alternate as in the quad levels.
            const uint32_t node = tid >> 4, w = tid & 15u, j = w >> 2;
            if (node < half) {
                const uint32_t *src = buf + base + 2 * node;
                const uint32_t ml = hashx::message(src[j * SMI_TOP_MAX], src[(4 + j) * SMI_TOP_MAX], row);
                const uint32_t mr = hashx::message(src[j * SMI_TOP_MAX + 1], src[(4 + j) * SMI_TOP_MAX + 1], row);
                const uint32_t x = hashx::node_hash(ml, mr, row);
                uint8_t *dst = (uint8_t *)(nodes + 2 * (level_offset(n, lvl) + (first >> (lvl - lvl_in)) + node));
                dst[w] = (uint8_t)x;
                dst[16 + w] = (uint8_t)(x >> 16);
                uint8_t *nb = (uint8_t *)(buf + (base ^ (SMI_TOP_MAX / 2)) + node);
                nb[4 * (j * SMI_TOP_MAX) + (w & 3u)] = (uint8_t)x;
                nb[4 * ((4 + j) * SMI_TOP_MAX) + (w & 3u)] = (uint8_t)(x >> 16);
            }
            base ^= SMI_TOP_MAX / 2;
            __syncthreads();
            continue;
        }
        if (SMI_TOP_QUAD && 4 * half <= SMI_TOP_THREADS) {
            // few nodes: a level is one node-hash latency, so each hash is spread over a quad of lanes
            // (hash_quad.h); lane q of the quad ends up with digest words q and q+4
            // These levels hold at most 512 digests, so they alternate between slots [0, 512) and
            // [1024, 1536) of each word row: one barrier per level instead of two.
            const uint32_t node = tid >> 2;
            if (node < half) {
                uint32_t l[8], r[8], lo, hi;
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    l[w] = buf[w * SMI_TOP_MAX + base + 2 * node];
                    r[w] = buf[w * SMI_TOP_MAX + base + 2 * node + 1];
                }
                hashq::node_hash(l, r, lane, lo, hi);
                uint32_t *dst = (uint32_t *)(nodes + 2 * (level_offset(n, lvl) + (first >> (lvl - lvl_in)) + node));
                dst[lane.q] = lo;
                dst[4 + lane.q] = hi;
                buf[lane.q * SMI_TOP_MAX + (base ^ (SMI_TOP_MAX / 2)) + node] = lo;
                buf[(4 + lane.q) * SMI_TOP_MAX + (base ^ (SMI_TOP_MAX / 2)) + node] = hi;
            }
            base ^= SMI_TOP_MAX / 2;
            __syncthreads();
            continue;
        }
        if (tid < half) {
            uint32_t l[8], r[8];
#pragma unroll
            for (int w = 0; w < 8; w++) {
                l[w] = buf[w * SMI_TOP_MAX + 2 * tid];
                r[w] = buf[w * SMI_TOP_MAX + 2 * tid + 1];
            }
            hashc::node_hash<SMI_TOP_UNROLL>(l, r, d);
            uint4 *dst = nodes + 2 * (level_offset(n, lvl) + (first >> (lvl - lvl_in)) + tid);
            dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
            dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
        }
        __syncthreads();   // every pair of this level has been read
        if (tid < half) {
#pragma unroll
            for (int w = 0; w < 8; w++) buf[w * SMI_TOP_MAX + tid] = d[w];
        }
        __syncthreads();
    }
    return base;
}

// the Fiat-Shamir round of the root by the launch that produced it (TopHook)
// (by the first sixteen lanes, one state word each: hash_hex.h)
__device__ __forceinline__ void top_finish(const TopHook &hook, const uint32_t *buf, uint32_t base) {
    if (hook.fs_words && threadIdx.x < 16 && gridDim.x == 1 && blockIdx.y == 0) {   // the root sits in slot `base` of every word row
        const hashx::Lane row = hashx::make_lane(threadIdx.x);
        const uint32_t j = threadIdx.x >> 2;
        hashx::fs_absorb_root(hook.fs_words, hashx::message(buf[j * SMI_TOP_MAX + base], buf[(4 + j) * SMI_TOP_MAX + base], row), row,
                              hook.proof_slot, hook.alpha_out);
    }
}
// one tree whose leaves are the fold of the round before (LeafSrc, LEAF_FOLD), chunk by chunk
__global__ __launch_bounds__(SMI_TOP_THREADS) void merkle_top_fold_kernel(uint4 *nodes, size_t n, uint32_t chunk, TopHook hook, const LeafSrc src) {
    __shared__ uint32_t buf[8 * SMI_TOP_MAX];
    const uint32_t base = top_chunk<true, true>(src.cw_out, nodes, n, 0, chunk, (size_t)blockIdx.x * chunk, 0, 0, buf, &src);
    top_finish(hook, buf, base);
}
template <bool FROM_ELEMS>
__global__ __launch_bounds__(SMI_TOP_THREADS) void merkle_top_kernel(const uint32_t *__restrict__ elems, uint4 *nodes, size_t n,
                                                                      uint32_t lvl_in, uint32_t chunk, size_t elem_stride,
                                                                      size_t node_stride, uint32_t row_cols, size_t row_stride,
                                                                      TopHook hook) {
    __shared__ uint32_t buf[8 * SMI_TOP_MAX];
    elems += (size_t)blockIdx.y * elem_stride;
    nodes += (size_t)blockIdx.y * node_stride;
    const uint32_t base = top_chunk<FROM_ELEMS>(elems, nodes, n, lvl_in, chunk, (size_t)blockIdx.x * chunk, row_cols, row_stride, buf);
    top_finish(hook, buf, base);
}

// The tail of Fri::commit (reference src/fri.rs:116-148) in ONE launch: once a codeword has at most
// SMI_TOP_MAX elements a single workgroup runs every remaining round -- leaf digests and the whole
// tree (current level in LDS), the Fiat-Shamir round of its root by lane 0, the fold into the next
// codeword -- with workgroup barriers where the host loop has kernel boundaries.  Trees, codewords,
// root records and challenges land where the per-round kernels put them (the query phase reads them).
__global__ __launch_bounds__(SMI_TOP_THREADS) void fri_tail_kernel(const FriTailArgs a) {
    __shared__ uint32_t buf[8 * SMI_TOP_MAX];
    const uint32_t tid = threadIdx.x;
    if (a.pre_lo) {   // the fold into the first codeword of the tail (otherwise a launch of its own before this one)
        const uint32_t ah_m = fold_alpha_half(*a.pre_alpha, a.inv2_m, a.F);
        uint32_t *cw0 = const_cast<uint32_t *>(a.r[0].cw);
        for (uint32_t i = tid; i < a.r[0].len; i += SMI_TOP_THREADS) cw0[i] = fold_element(a.pre_lo[i], a.pre_hi[i], i, ah_m, a.inv2_m, a.pre_S, a.F);
        __threadfence_block();
        __syncthreads();
    }
    for (uint32_t k = 0; k < a.n_rounds; k++) {
        const FriTailRound R = a.r[k];
        const uint32_t base = top_chunk<true>(R.cw, (uint4 *)R.nodes, R.len, 0, R.len, 0, 0, 0, buf);
        if (tid < 16) {
            const hashx::Lane row = hashx::make_lane(tid);
            const uint32_t j = tid >> 2;
            hashx::fs_absorb_root(a.fs_words, hashx::message(buf[j * SMI_TOP_MAX + base], buf[(4 + j) * SMI_TOP_MAX + base], row), row,
                                  R.proof_slot, R.alpha_out);   // no challenge after the last root
            __threadfence_block();
        }
        __syncthreads();   // the challenge is in memory; buf may be overwritten by the next round
        if (!R.next) break;
        const uint32_t ah_m = fold_alpha_half(*(volatile const uint64_t *)R.alpha_out, a.inv2_m, a.F);
        const uint32_t half = R.len >> 1;
        for (uint32_t i = tid; i < half; i += SMI_TOP_THREADS) R.next[i] = fold_element(R.cw[i], R.cw[i + half], i, ah_m, a.inv2_m, R.S, a.F);
        __threadfence_block();
        __syncthreads();   // the next codeword is complete before its leaves are hashed
    }
}
int launch_fri_tail(smi_ctx *ctx, const FriTailArgs &a) {
    if (!a.n_rounds) return SMI_OK;
    if (a.n_rounds > SMI_FRI_TAIL_MAX_ROUNDS || a.r[0].len > SMI_TOP_MAX) return smi_fail(ctx, SMI_ERR_BAD_ARG, "fri tail: too long");
    double tail_mixes = 0.0;   // per round: len leaves, len - 1 nodes (the Fiat-Shamir hashes are not counted)
    for (uint32_t k = 0; k < a.n_rounds; k++) tail_mixes += 9.0 * a.r[k].len + 10.0 * ((double)a.r[k].len - 1.0);
    ProfScope ps(ctx, "fri_tail_kernel", 0.0, tail_mixes);
    fri_tail_kernel<<<1, SMI_TOP_THREADS, 0, ctx->stream>>>(a);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

// digests of rows of any width (the fused kernels above take rows of up to 4 columns)
__global__ __launch_bounds__(SMI_HASH_THREADS) void row_hash_kernel(const uint32_t *__restrict__ cols, uint32_t n_cols, size_t stride,
                                                                      uint4 *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * SMI_HASH_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t row[SMI_ROW_MAX], d[8];
    for (uint32_t c = 0; c < n_cols; c++) row[c] = cols[c * stride + i];
    hashc::row_hash(row, (int)n_cols, d);
    out[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
    out[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// digests only (Hash::from_field_elements per element)
__global__ __launch_bounds__(SMI_HASH_THREADS) void leaf_hash_kernel(const uint32_t *__restrict__ elems, uint4 *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * SMI_HASH_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t d[8];
    hashc::leaf_hash(elems[i], d);
    out[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
    out[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// out[i] = combine(in[2i], in[2i+1])
__global__ __launch_bounds__(SMI_HASH_THREADS) void combine_kernel(const uint4 *__restrict__ in, uint4 *out, size_t n_pairs) {
    const size_t i = (size_t)blockIdx.x * SMI_HASH_THREADS + threadIdx.x;
    if (i >= n_pairs) return;
    const uint4 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], e = in[4 * i + 3];
    const uint32_t l[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, r[8] = {c.x, c.y, c.z, c.w, e.x, e.y, e.z, e.w};
    uint32_t d[8];
    hashc::node_hash(l, r, d);
    out[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
    out[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// MerkleTree::verify (src/merkle.rs:82-96) for k (leaf, index, path) triples against one root
__global__ __launch_bounds__(SMI_HASH_THREADS) void verify_paths_kernel(const uint4 *__restrict__ leaves, const uint64_t *__restrict__ indices,
                                                                          const uint4 *__restrict__ paths, size_t k, uint32_t depth,
                                                                          const uint4 *__restrict__ root, uint8_t *ok) {
    const size_t i = (size_t)blockIdx.x * SMI_HASH_THREADS + threadIdx.x;
    if (i >= k) return;
    uint4 a = leaves[2 * i], b = leaves[2 * i + 1];
    uint32_t cur[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint64_t idx = indices[i];
    for (uint32_t d = 0; d < depth; d++) {
        a = paths[2 * (i * depth + d)];
        b = paths[2 * (i * depth + d) + 1];
        const uint32_t sib[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t nxt[8];
        if (idx % 2 == 0) hashc::node_hash(cur, sib, nxt);
        else hashc::node_hash(sib, cur, nxt);
#pragma unroll
        for (int w = 0; w < 8; w++) cur[w] = nxt[w];
        idx /= 2;
    }
    a = root[0];
    b = root[1];
    ok[i] = cur[0] == a.x && cur[1] == a.y && cur[2] == a.z && cur[3] == a.w && cur[4] == b.x && cur[5] == b.y && cur[6] == b.z &&
            cur[7] == b.w;
}

// Hash::from_bytes of one message, single lane (src/hash.rs:7-30)
__global__ void hash_bytes_kernel(const uint8_t *msg, size_t len, uint32_t *out) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t d[8];
    hashc::hash_bytes(msg, len, d);
    for (int i = 0; i < 8; i++) out[i] = d[i];
}

// n messages of the same length, one lane each (index sampling hashes seed || counter for a run
// of counters, src/fri.rs:176-213)
__global__ __launch_bounds__(64) void hash_bytes_batch_kernel(const uint8_t *msgs, size_t n, size_t len, uint32_t *out) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t d[8];
    hashc::hash_bytes(msgs + i * len, len, d);
    for (int k = 0; k < 8; k++) out[8 * i + k] = d[k];
}

// smi_ctx_mix_probe: the bare permutation, two hashes per lane (State2), nothing else -- the ceiling the Merkle
// kernels are reported against (bench.py prove_roofline).  The state is seeded per lane and folded into one
// word at the end so that nothing is optimised away.
__global__ __launch_bounds__(SMI_HASH_THREADS) void mix_probe_kernel(uint32_t *out, uint32_t mixes) {
    hashc::State2 st;
#pragma unroll
    for (int w = 0; w < 32; w++) st.s[w] = (threadIdx.x * 2654435761u + (uint32_t)w * 40503u + blockIdx.x) & 0x00FF00FFu;
    const hashc::MixK K = hashc::mix_consts();
#pragma unroll 1
    for (uint32_t i = 0; i < mixes; i++) hashc::mix2_t<true>(st, K);
    uint32_t x = 0;
#pragma unroll
    for (int w = 0; w < 32; w++) x ^= st.s[w];
    out[(size_t)blockIdx.x * SMI_HASH_THREADS + threadIdx.x] = x;
}
int smi_ctx_mix_probe(smi_ctx *ctx, uint32_t mixes, double *mixes_per_s) {
    if (!ctx || !mixes_per_s || !mixes) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    const uint32_t blocks = (uint32_t)ctx->num_cus * 40;          // 40 workgroups per CU: several full waves of the chip
    void *d_out = nullptr;
    SMI_TRY(ctx_tmp(ctx, 3, (size_t)blocks * SMI_HASH_THREADS * 4, &d_out));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return smi_fail(ctx, SMI_ERR_HIP, "hipEventCreate"); }
    mix_probe_kernel<<<blocks, SMI_HASH_THREADS, 0, ctx->stream>>>((uint32_t *)d_out, mixes);   // warm-up (clocks, code)
    (void)hipEventRecord(e0, ctx->stream);
    mix_probe_kernel<<<blocks, SMI_HASH_THREADS, 0, ctx->stream>>>((uint32_t *)d_out, mixes);
    (void)hipEventRecord(e1, ctx->stream);
    const hipError_t err = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess || hipGetLastError() != hipSuccess || ms <= 0.f) return smi_fail(ctx, SMI_ERR_HIP, "mix probe");
    *mixes_per_s = 2.0 * (double)blocks * SMI_HASH_THREADS * (double)mixes / ((double)ms * 1e-3);
    return SMI_OK;
}

// ------------------------------------------------------------------------- launches
static inline uint32_t blocks_for(size_t threads) { return (uint32_t)((threads + SMI_HASH_THREADS - 1) / SMI_HASH_THREADS); }

int launch_leaf_hash(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_digests) {
    if (!n) return SMI_OK;
    leaf_hash_kernel<<<blocks_for(n), SMI_HASH_THREADS, 0, ctx->stream>>>(d_elems, (uint4 *)d_digests, n);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_combine(smi_ctx *ctx, const uint8_t *d_in, size_t n_pairs, uint8_t *d_out) {
    if (!n_pairs) return SMI_OK;
    combine_kernel<<<blocks_for(n_pairs), SMI_HASH_THREADS, 0, ctx->stream>>>((const uint4 *)d_in, (uint4 *)d_out, n_pairs);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_verify_paths(smi_ctx *ctx, const uint8_t *d_leaves, const uint64_t *d_idx, const uint8_t *d_paths, size_t k, uint32_t depth,
                        const uint8_t *d_root, uint8_t *d_ok) {
    if (!k) return SMI_OK;
    verify_paths_kernel<<<blocks_for(k), SMI_HASH_THREADS, 0, ctx->stream>>>((const uint4 *)d_leaves, d_idx, (const uint4 *)d_paths, k, depth,
                                                                              (const uint4 *)d_root, d_ok);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}
int launch_hash_bytes(smi_ctx *ctx, const uint8_t *d_msg, size_t len, uint32_t *d_out) {
    hash_bytes_kernel<<<1, 64, 0, ctx->stream>>>(d_msg, len, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

int launch_hash_bytes_batch(smi_ctx *ctx, const uint8_t *d_msgs, size_t n, size_t len, uint32_t *d_out) {
    if (!n) return SMI_OK;
    hash_bytes_batch_kernel<<<(uint32_t)((n + 63) / 64), 64, 0, ctx->stream>>>(d_msgs, n, len, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SMI_OK;
}

static uint32_t log2_floor(size_t n) {
    uint32_t l = 0;
    while ((n >> l) > 1) l++;
    return l;
}

// Builds levels (lvl_from, log2 n] of the tree in d_nodes; if d_elems != nullptr level 0 is
// hashed from the codeword first (fused with the bottom levels).  n must be a power of two.
int launch_merkle_batch(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t n_trees, size_t elem_stride,
                        size_t node_stride_bytes, uint32_t row_cols = 0, size_t row_stride = 0);
static int launch_merkle_impl(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t n_trees, size_t elem_stride,
                              size_t node_stride_bytes, uint32_t row_cols, size_t row_stride, const TopHook *hook, bool *hook_done,
                              const LeafSrc *src = nullptr);
int launch_merkle(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes) {
    return launch_merkle_batch(ctx, d_elems, n, d_nodes, 1, 0, 0);
}
// one tree, with the Fiat-Shamir round of its root run by the launch that produces it when that is the
// chunk kernel (*done tells the caller whether it was)
int launch_merkle_fs(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t *fs_words, uint8_t *proof_slot,
                     uint64_t *alpha_out, bool *done) {
    const TopHook hook{fs_words, proof_slot, alpha_out};
    *done = false;
    return launch_merkle_impl(ctx, d_elems, n, d_nodes, 1, 0, 0, 0, 0, &hook, done);
}
// The planner's rule, restated: with more than 2048 * TOP_BLOCKS leaves the first launch of a single tree is the
// four-leaves-per-lane kernel (the chunk kernel takes over below that), and that kernel can compute its leaves (LeafSrc).
static size_t merkle_top_blocks() {
    static const size_t v = [] { const char *e = getenv("SMI_MERKLE_TOP_BLOCKS"); return (size_t)(e ? atoi(e) : 256); }();
    return v;
}
// A tree that starts from codeword elements goes to the chunk kernel only once it has at most this many leaves: a chunk's
// first level is then leaf hashing, nine mixes per leaf by single lanes, which the four-leaves-per-lane kernel does at twice
// the rate even when it fills a quarter of the chip (tuning knob SMI_MERKLE_ELEMS_LOG, log2)
static size_t merkle_elems_max() {
    static const size_t v = [] { const char *e = getenv("SMI_MERKLE_ELEMS_LOG"); return (size_t)1 << (e ? atoi(e) : 19); }();
    const size_t by_blocks = (size_t)SMI_TOP_MAX * merkle_top_blocks();
    return v < by_blocks ? v : by_blocks;
}
bool merkle_fuses_leaf_source(size_t n) {
    static const bool off = (getenv("SMI_MERKLE_FUSE") && atoi(getenv("SMI_MERKLE_FUSE")) == 0) ||
                            (getenv("SMI_MERKLE_GENERIC") && atoi(getenv("SMI_MERKLE_GENERIC"))) ||
                            (getenv("SMI_MERKLE_K") && atoi(getenv("SMI_MERKLE_K")) != 2);
    return !off && n >= 8 && (n & (n - 1)) == 0 && n > merkle_elems_max();
}
// ... and with at most that many (and at least two) the first launch is the chunk kernel, which can fold as it reads
bool merkle_chunks_fold(size_t n) {
    static const bool off = getenv("SMI_MERKLE_FUSE") && atoi(getenv("SMI_MERKLE_FUSE")) == 0;
    return !off && n >= 2 && (n & (n - 1)) == 0 && n <= merkle_elems_max();
}
// one tree whose leaves are computed by the launch that hashes them (src.cw_out receives the codeword)
int launch_merkle_src_fs(smi_ctx *ctx, const LeafSrc &src, size_t n, uint8_t *d_nodes, uint32_t *fs_words, uint8_t *proof_slot,
                         uint64_t *alpha_out, bool *done) {
    const bool ok = merkle_fuses_leaf_source(n) || (src.kind == LEAF_FOLD && merkle_chunks_fold(n));
    if (!ok || !src.cw_out || (src.kind == LEAF_COMBINE && (!src.n_cols || src.n_cols > SMI_LEAF_COMBINE_MAX)))
        return smi_fail(ctx, SMI_ERR_BAD_ARG, "merkle: this tree cannot take a computed leaf source");
    const TopHook hook{fs_words, proof_slot, alpha_out};
    *done = false;
    return launch_merkle_impl(ctx, src.cw_out, n, d_nodes, 1, 0, 0, 0, 0, &hook, done, &src);
}
// one tree whose leaf i hashes row i of n_cols columns (column c at d_cols + c*col_stride)
int launch_merkle_rows(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t col_stride, size_t n, uint8_t *d_nodes) {
    if (!n_cols || n_cols > SMI_ROW_MAX) return smi_fail(ctx, SMI_ERR_BAD_ARG, "row leaves: 1..64 columns");
    if (n_cols <= 4) return launch_merkle_batch(ctx, d_cols, n, d_nodes, 1, 0, 0, n_cols, col_stride);
    row_hash_kernel<<<blocks_for(n), SMI_HASH_THREADS, 0, ctx->stream>>>(d_cols, n_cols, col_stride, (uint4 *)d_nodes, n);
    HIP_TRY(ctx, hipGetLastError());
    return launch_merkle_batch(ctx, nullptr, n, d_nodes, 1, 0, 0);
}
// n_trees equally sized trees in one set of launches: tree y reads d_elems + y*elem_stride and writes
// d_nodes + y*node_stride_bytes.  The small upper levels of all trees share their launch latency.
int launch_merkle_batch(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t n_trees, size_t elem_stride,
                        size_t node_stride_bytes, uint32_t row_cols, size_t row_stride) {
    return launch_merkle_impl(ctx, d_elems, n, d_nodes, n_trees, elem_stride, node_stride_bytes, row_cols, row_stride, nullptr, nullptr);
}
static int launch_merkle_impl(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t n_trees, size_t elem_stride,
                              size_t node_stride_bytes, uint32_t row_cols, size_t row_stride, const TopHook *hook, bool *hook_done,
                              const LeafSrc *src) {
    if (!n_trees) return SMI_OK;
    LeafSrc none;
    memset(&none, 0, sizeof none);
    const size_t node_stride = node_stride_bytes / 16;
    const uint32_t depth = log2_floor(n);
    uint4 *nodes = (uint4 *)d_nodes;
    uint32_t lvl = 0;
    size_t count = n;
    bool from_elems = d_elems != nullptr;
    static const uint32_t KMAX = [] {  // levels fused per launch (tuning knob; LDS stash = 8 KB << K)
        const char *e = getenv("SMI_MERKLE_K");
        const int k = e ? atoi(e) : 2;
        return (uint32_t)(k < 1 ? 1 : (k > 3 ? 3 : k));
    }();
    // chunk workgroups per launch below which the chunk kernel takes over (one per CU; tuning knob)
    const size_t TOP_BLOCKS = merkle_top_blocks();
    if (from_elems && depth == 0 && !row_cols) {
        for (uint32_t y = 0; y < n_trees; y++) SMI_TRY(launch_leaf_hash(ctx, d_elems + y * elem_stride, 1, d_nodes + y * node_stride_bytes));
        return SMI_OK;
    }
    while (lvl < depth || from_elems) {
        // Once what is left fits the chip as one wave of chunk workgroups, the per-level latency of
        // the chunk kernel beats two-level launches: up to SMI_TOP_MAX digests (11 levels) per launch.
        // Chunks are as small as one workgroup per CU allows (not below 64): the widest levels of a chunk
        // are throughput on a single CU, so 256 chunks of 256 digests and then their 256 roots finish a
        // 2^16-leaf tree in 42 us where 32 chunks of 2048 took 65.  Up to 512 digests stay one chunk.
        // (Both bounds are tuning knobs; tools/sweep_merkle_chunks.sh: flat within 3 % from 8 to 64.)
        static const size_t SINGLE_MAX = [] { const char *e = getenv("SMI_MERKLE_SINGLE"); return (size_t)(e ? atoi(e) : 512); }();
        static const size_t MIN_CHUNK = [] { const char *e = getenv("SMI_MERKLE_MINCHUNK"); return (size_t)(e ? atoi(e) : 64); }();
        size_t chunk = count;
        if (count > SINGLE_MAX) {
            chunk = MIN_CHUNK;
            while (chunk < SMI_TOP_MAX && (count / chunk) * n_trees > TOP_BLOCKS) chunk <<= 1;
        }
        const size_t n_chunks = count / chunk;
        if (chunk <= SMI_TOP_MAX && n_chunks * n_trees <= TOP_BLOCKS && (!from_elems || row_cols || count <= merkle_elems_max())) {
            const double hashed = (from_elems ? 2.0 * (double)count : (double)count) - (double)n_chunks;
            // mix_state evaluations: 9 per single-element leaf (one more per extra 32-byte chunk of a row), 10 per node
            const double leaf_mixes = 8.0 + (row_cols ? (double)((row_cols + 3) / 4) : 1.0);
            const double mixes = (from_elems ? leaf_mixes * (double)count : 0.0) + 10.0 * ((double)count - (double)n_chunks);
            ProfScope ps(ctx, "merkle_top_kernel", ((from_elems ? 4.0 * (row_cols ? row_cols : 1) : 32.0) * (double)count + 32.0 * hashed) * n_trees,
                         mixes * n_trees);
            const dim3 grid((uint32_t)n_chunks, n_trees);
            TopHook h{nullptr, nullptr, nullptr};
            if (hook && n_chunks == 1 && n_trees == 1) {   // this launch ends with the root
                h = *hook;
                *hook_done = true;
            }
            if (from_elems && src) {
                if (src->kind != LEAF_FOLD || n_trees != 1 || row_cols) return smi_fail(ctx, SMI_ERR_BAD_ARG, "merkle: the chunk kernel computes folded leaves only");
                merkle_top_fold_kernel<<<grid, SMI_TOP_THREADS, 0, ctx->stream>>>(nodes, n, (uint32_t)chunk, h, *src);
            } else if (from_elems)
                merkle_top_kernel<true><<<grid, SMI_TOP_THREADS, 0, ctx->stream>>>(d_elems, nodes, n, 0, (uint32_t)chunk, elem_stride, node_stride, row_cols, row_stride, h);
            else
                merkle_top_kernel<false><<<grid, SMI_TOP_THREADS, 0, ctx->stream>>>(nullptr, nodes, n, lvl, (uint32_t)chunk, 0, node_stride, 0, 0, h);
            HIP_TRY(ctx, hipGetLastError());
            from_elems = false;
            uint32_t up = 0;
            while (((size_t)1 << up) < chunk) up++;
            lvl += up;
            count = n_chunks;
            continue;
        }
        uint32_t K = depth - lvl < KMAX ? depth - lvl : KMAX;
        const size_t threads = count >> K;
        const size_t lds = (size_t)(8u << K) * SMI_HASH_THREADS * sizeof(uint32_t);
        // algorithmic bytes: inputs read once (4 B elements or 32 B digests), every produced digest written once
        const double produced = from_elems ? (double)count * 2.0 - (double)(count >> K) : (double)count - (double)(count >> K);
        const double sub_mixes = (from_elems ? (8.0 + (row_cols ? (double)((row_cols + 3) / 4) : 1.0)) * (double)count : 0.0) +
                                 10.0 * ((double)count - (double)(count >> K));
        ProfScope ps(ctx, from_elems ? "merkle_sub_kernel<leaves>" : "merkle_sub_kernel<digests>",
                     ((from_elems ? 4.0 * (row_cols ? row_cols : 1) : 32.0) * (double)count + 32.0 * produced) * n_trees, sub_mixes * n_trees);
        const dim3 grid(blocks_for(threads), n_trees);
        // the hot shapes (element leaves or digests, two levels per lane) have instantiations of their own
        static const bool generic_only = getenv("SMI_MERKLE_GENERIC") && atoi(getenv("SMI_MERKLE_GENERIC"));
        if (from_elems && row_cols)
            merkle_sub_kernel<true, 0, true><<<grid, SMI_HASH_THREADS, lds, ctx->stream>>>(d_elems, nodes, n, 0, count, K, elem_stride, node_stride, row_cols, row_stride, none);
        else if (from_elems && K == 2 && !generic_only && src && src->kind == LEAF_FOLD)   // leaves computed on the fly (LeafSrc)
            merkle_sub_kernel<true, 2, false, LEAF_FOLD><<<grid, SMI_HASH_THREADS, (size_t)16 * SMI_HASH_THREADS * sizeof(uint32_t), ctx->stream>>>(d_elems, nodes, n, 0, count, K, elem_stride, node_stride, 0, 0, *src);
        else if (from_elems && K == 2 && !generic_only && src && src->kind == LEAF_COMBINE)
            merkle_sub_kernel<true, 2, false, LEAF_COMBINE><<<grid, SMI_HASH_THREADS, (size_t)16 * SMI_HASH_THREADS * sizeof(uint32_t), ctx->stream>>>(d_elems, nodes, n, 0, count, K, elem_stride, node_stride, 0, 0, *src);
        else if (src && from_elems)
            return smi_fail(ctx, SMI_ERR_BAD_ARG, "merkle: computed leaves need the four-leaves-per-lane kernel");
        else if (from_elems && K == 2 && !generic_only)   // its stash holds two digests per lane
            merkle_sub_kernel<true, 2, false><<<grid, SMI_HASH_THREADS, (size_t)16 * SMI_HASH_THREADS * sizeof(uint32_t), ctx->stream>>>(d_elems, nodes, n, 0, count, K, elem_stride, node_stride, 0, 0, none);
        else if (from_elems)
            merkle_sub_kernel<true, 0, false><<<grid, SMI_HASH_THREADS, lds, ctx->stream>>>(d_elems, nodes, n, 0, count, K, elem_stride, node_stride, 0, 0, none);
        else if (K == 2 && !generic_only)                 // no stash at all
            merkle_sub_kernel<false, 2, false><<<grid, SMI_HASH_THREADS, 0, ctx->stream>>>(nullptr, nodes, n, lvl, count, K, 0, node_stride, 0, 0, none);
        else
            merkle_sub_kernel<false, 0, false><<<grid, SMI_HASH_THREADS, lds, ctx->stream>>>(nullptr, nodes, n, lvl, count, K, 0, node_stride, 0, 0, none);
        HIP_TRY(ctx, hipGetLastError());
        from_elems = false;
        lvl += K;
        count >>= K;
    }
    return SMI_OK;
}
