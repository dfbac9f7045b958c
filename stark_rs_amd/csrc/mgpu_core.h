// mgpu_core.h -- device-side pieces of the multi-GPU prover that the HIP kernels (mgpu.hip) and
// the CPU emulator of the non-GPU tests (emu_mgpu.cpp) share: how an opening of a sharded
// codeword is served and written into the serialized proof.
//
// One codeword of length `len` is split in contiguous blocks of `blk` elements, rank g holding
// [g*blk, (g+1)*blk) and the complete Merkle subtree over it (a power-of-two aligned leaf block is
// a subtree of MerkleTree::new, reference src/merkle.rs:21-31); the G sub-roots are all-gathered
// and every rank holds the log2 G levels above them (`top`).  An opening (value + authentication
// path, Fri::query src/fri.rs:215-248, MerkleTree::open src/merkle.rs:67-80) is written by the
// rank that owns the leaf; everything replicated (tags, lengths, roots, the last codeword) by
// rank 0; the proof buffer starts zeroed on every rank, so a byte-wise sum over the ranks
// (all-reduce) assembles the reference's serialization (src/stream.rs:35-64) on all of them.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "field.h"

struct MgSide {            // one round's codeword and tree as THIS rank holds them
    const uint32_t *cw;    // the local block (the whole codeword when the round is replicated)
    const uint8_t *nodes;  // all levels of the tree over the local block, level 0 first
    const uint8_t *top;    // all levels of the tree over the G sub-roots (sharded rounds), else unused
    uint64_t len;          // length of the round's whole codeword
    uint64_t blk;          // length of the local block (== len when replicated)
    uint32_t depth_local;  // log2 blk
    uint32_t depth_top;    // log2 G for a sharded round, 0 otherwise
};
struct MgLayer {           // one FRI layer of the query phase (src/fri.rs:280-308)
    MgSide cur, next;
    uint64_t off_triples;  // byte offset of this layer's first FieldElements triple
    uint64_t off_paths;    // byte offset of this layer's first MerklePath
};

SMI_HD void mg_put_u64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

// Value slot (8 bytes LE) and path (tag 3, u64 count, 32-byte digests) of leaf `index` of `side`.
// lane / n_lanes: the caller's cooperative threads (n_lanes >= 32 on the device, 1 in the emulator).
SMI_HD void mg_open_write(const MgSide &side, uint64_t index, int rank, uint8_t *value_slot, uint8_t *path_slot, uint32_t lane,
                          uint32_t n_lanes) {
    const uint32_t depth = side.depth_local + side.depth_top;
    if (rank == 0 && lane == 0) {            // replicated bytes: rank 0
        path_slot[0] = 3;
        mg_put_u64(path_slot + 1, depth);
    }
    const uint64_t owner = side.depth_top ? index / side.blk : 0;
    if ((uint64_t)rank != owner) return;
    const uint64_t local = index - owner * side.blk;
    if (lane == 0) mg_put_u64(value_slot, side.cw[local]);
    uint64_t idx = local, lvl_off = 0, n = side.blk;
    for (uint32_t l = 0; l < side.depth_local; l++) {          // inside the resident subtree
        const uint64_t sib = idx ^ 1;
        for (uint32_t b = lane; b < 32; b += n_lanes) path_slot[9 + 32 * l + b] = side.nodes[(lvl_off + sib) * 32 + b];
        idx >>= 1;
        lvl_off += n;
        n >>= 1;
    }
    idx = owner;
    lvl_off = 0;
    n = (uint64_t)1 << side.depth_top;
    for (uint32_t l = 0; l < side.depth_top; l++) {            // above the sub-roots (replicated levels)
        const uint64_t sib = idx ^ 1;
        for (uint32_t b = lane; b < 32; b += n_lanes)
            path_slot[9 + 32 * (side.depth_local + l) + b] = side.top[(lvl_off + sib) * 32 + b];
        idx >>= 1;
        lvl_off += n;
        n >>= 1;
    }
}

// Test s of layer L: the triple (a, b, c) and the three paths, src/fri.rs:229-243.
SMI_HD void mg_query_write(const MgLayer &L, uint64_t top_index, uint32_t s, int rank, uint8_t *proof, uint32_t lane, uint32_t n_lanes) {
    const uint64_t half = L.cur.len / 2;
    const uint64_t c = top_index % half;     // indices folded layer by layer: (x % a) % b == x % b for b | a
    uint8_t *tr = proof + L.off_triples + (uint64_t)s * 33;
    if (rank == 0 && lane == 0) {
        tr[0] = 2;
        mg_put_u64(tr + 1, 3);
    }
    const uint64_t pa = 9 + 32ull * (L.cur.depth_local + L.cur.depth_top), pc = 9 + 32ull * (L.next.depth_local + L.next.depth_top);
    uint8_t *pp = proof + L.off_paths + (uint64_t)s * (2 * pa + pc);
    mg_open_write(L.cur, c, rank, tr + 9, pp, lane, n_lanes);
    mg_open_write(L.cur, c + half, rank, tr + 17, pp + pa, lane, n_lanes);
    mg_open_write(L.next, c, rank, tr + 25, pp + 2 * pa, lane, n_lanes);
}

// Column openings of the build-defined composition (smi_stark_cfg.open_columns): what ties the codeword FRI
// proves low-degree to the W committed columns.  For test s: a = top mod N/2, b = a + N/2;
//   rows : t x 2 records  tag 2 | u64 W | W x u64            (row a, then row b)
//   paths: t x W x 2 records  tag 3 | u64 depth | depth x 32  (column c at a, then at b)
// written like the FRI openings: values and digests by the rank that owns the leaf, tags by rank 0.
SMI_HD uint64_t mg_column_open_bytes(uint32_t W, uint32_t t, uint32_t depth) {
    return (uint64_t)t * 2 * (9 + 8ull * W) + (uint64_t)t * W * 2 * (9 + 32ull * depth);
}
SMI_HD void mg_column_open_write(const MgSide *cols, uint32_t W, uint32_t c, uint64_t top_index, uint32_t s, uint32_t t, int rank,
                                 uint8_t *out, uint32_t lane, uint32_t n_lanes) {
    const MgSide &side = cols[c];
    const uint32_t depth = side.depth_local + side.depth_top;
    const uint64_t half = side.len / 2, a = top_index % half, rec = 9 + 8ull * W, prec = 9 + 32ull * depth;
    uint8_t *rows = out + (uint64_t)s * 2 * rec;
    uint8_t *paths = out + (uint64_t)t * 2 * rec + ((uint64_t)s * W + c) * 2 * prec;
    if (rank == 0 && c == 0 && lane == 0)
        for (int k = 0; k < 2; k++) {
            rows[k * rec] = 2;
            mg_put_u64(rows + k * rec + 1, W);
        }
    mg_open_write(side, a, rank, rows + 9 + 8ull * c, paths, lane, n_lanes);
    mg_open_write(side, a + half, rank, rows + rec + 9 + 8ull * c, paths + prec, lane, n_lanes);
}
