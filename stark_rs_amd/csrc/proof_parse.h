// proof_parse.h -- ProofStream::deserialize (reference src/stream.rs:66-168) over the serialized bytes, host code.
// The one place where the library reads bytes it did not write (a proof handed to smi_fri_verify /
// smi_stark_verify): kept free of HIP so that the CPU build (csrc/emu.cpp) compiles the same code for the
// sanitizer and fuzz tests (tests/test_proof_parse.py, `make -C stark_rs_amd asan`).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace proofp {
struct Obj {           // one ProofObject (src/stream.rs:4-9), pointing into the proof bytes
    int tag;           // 0 MerkleRoot, 1 FieldElement, 2 FieldElements, 3 MerklePath
    const uint8_t *p;  // payload: 32 bytes | 8 bytes | count x 8 | count x 32
    size_t count;
};
inline uint64_t get_u64(const uint8_t *b) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)b[i] << (8 * i);
    return v;
}
// An object whose header or payload is cut short is dropped or shortened exactly as the reference's loops do;
// an unknown tag ends the stream.  Stops after max_objs objects; *end = the byte offset reached.  Never reads
// at or beyond b + n: every length taken from the bytes is clamped to what is left.
inline std::vector<Obj> parse(const uint8_t *b, size_t n, size_t max_objs, size_t *end) {
    std::vector<Obj> out;
    size_t i = 0;
    while (i < n && out.size() < max_objs) {
        const uint8_t tag = b[i];
        i++;
        if (tag == 0) {
            if (n - i >= 32) { out.push_back(Obj{0, b + i, 1}); i += 32; }
        } else if (tag == 1) {
            if (n - i >= 8) { out.push_back(Obj{1, b + i, 1}); i += 8; }
        } else if (tag == 2 || tag == 3) {
            if (n - i >= 8) {
                const uint64_t len = get_u64(b + i);
                i += 8;
                const size_t w = tag == 2 ? 8 : 32, avail = (n - i) / w, take = len < avail ? (size_t)len : avail;
                out.push_back(Obj{tag, b + i, take});
                i += take * w;
            }
        } else {
            i--;
            break;
        }
    }
    *end = i;
    return out;
}
}  // namespace proofp
