// verify.hip -- Fri::verify (reference src/fri.rs:313-504) behind the C ABI, and the verifier of the
// build-defined composition's column openings (smi_stark_cfg.open_columns).
//
// The reference's control flow runs on the host over the serialized ProofStream (src/stream.rs:66-168,
// with its leniency for truncated objects); the work that scales with the proof goes to the device
// in batches: every leaf hash (Hash::from_field_elements(&[v]) = the hash of v's 8 LE bytes,
// src/hash.rs:32-35 -- taken from the raw u64, so an unreduced value hashes as the reference hashes
// it), every authentication path (MerkleTree::verify, src/merkle.rs:82-96), the Merkle root of the
// last codeword, and the last layer's low-degree test as an inverse + forward NTT instead of the
// reference's O(L^3) Lagrange interpolation.  *accept is 1 where the reference returns
// true and 0 where it prints a reason and returns false (smi_last_error carries the reason); where
// the reference panics the status is that panic's code.
#include <string.h>

#include <vector>

#include "internal.h"
#include "proof_parse.h"

namespace {
using proofp::Obj;
using proofp::get_u64;
using proofp::parse;
bool pow2(uint64_t n) { return n && !(n & (n - 1)); }
uint32_t ilog2(uint64_t n) {
    uint32_t l = 0;
    while ((n >> l) > 1) l++;
    return l;
}
uint64_t mulm(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((unsigned __int128)a * b % p); }
// FiniteField::sub (src/ff.rs:154-160): `p + l - r` in u128, then `% p`; for an unreduced r > p + l a release build wraps
// mod 2^128 before the reduction (a debug build panics) -- the oracle restates the release behaviour and so does this
uint64_t subm(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((((unsigned __int128)p + a) - b) % p); }
uint64_t powm(uint64_t b, uint64_t e, uint64_t p) {
    uint64_t r = 1 % p;
    b %= p;
    while (e) {
        if (e & 1) r = mulm(r, b, p);
        b = mulm(b, b, p);
        e >>= 1;
    }
    return r;
}
int reject(smi_ctx *ctx, int *accept, const char *why) {
    *accept = 0;
    ctx->err = why;
    return SMI_OK;
}
// leaf digests of raw u64 values: Hash::from_bytes(v.to_le_bytes())
int leaf_digests(smi_ctx *ctx, const uint64_t *v, size_t n, std::vector<uint8_t> &out) {
    std::vector<uint8_t> msgs(8 * n);
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) msgs[8 * i + k] = (uint8_t)(v[i] >> (8 * k));
    out.resize(32 * n);
    return n ? smi_hash_bytes_batch(ctx, msgs.data(), n, 8, out.data()) : SMI_OK;
}
// FiatShamir::challenge over the transcript of the first k roots
int challenge_of(smi_ctx *ctx, const std::vector<uint8_t> &transcript, uint64_t *out) {
    uint8_t d[32];
    SMI_TRY(smi_hash_bytes(ctx, transcript.data(), transcript.size(), d));
    *out = get_u64(d);
    return SMI_OK;
}

// Fri::verify on objs[0..]; *used = objects consumed on acceptance
int fri_verify_objs(smi_ctx *ctx, const smi_fri_cfg &cfg, const std::vector<Obj> &objs, int *accept, std::vector<uint64_t> *top_out,
                    std::vector<uint64_t> *pv_idx, std::vector<uint64_t> *pv_val, std::vector<uint64_t> *layer0_ab, size_t *used) {
    const uint64_t p = ctx->fs.F.p, t = cfg.num_colinearity_tests, N = cfg.domain_length;
    uint64_t R = 0;
    smi_fri_num_rounds(&cfg, &R);
    size_t at = 0;
    auto pop = [&]() -> const Obj * { return at < objs.size() ? &objs[at++] : nullptr; };
    std::vector<uint8_t> transcript;
    std::vector<const uint8_t *> roots;
    std::vector<uint64_t> alphas;
    for (uint64_t r = 0; r < R; r++) {                                             // src/fri.rs:325-334
        const Obj *o = pop();
        if (!o || o->tag != 0) return reject(ctx, accept, "Failed to extract Merkle root");
        roots.push_back(o->p);
        transcript.insert(transcript.end(), o->p, o->p + 32);
        uint64_t a = 0;
        SMI_TRY(challenge_of(ctx, transcript, &a));
        alphas.push_back(a);
    }
    const Obj *lo = pop();                                                          // :337-342
    if (!lo || lo->tag != 2) return reject(ctx, accept, "Failed to extract last codeword");
    if (R == 0) return reject(ctx, accept, "No FRI roots extracted");               // :345-348
    const size_t n_last = lo->count;
    std::vector<uint64_t> last(n_last);
    for (size_t i = 0; i < n_last; i++) last[i] = get_u64(lo->p + 8 * i);
    if (n_last == 0) return smi_fail(ctx, SMI_ERR_EMPTY_LEAVES, nullptr);          // MerkleTree::new panics, :353
    if (!pow2(n_last)) return smi_fail(ctx, SMI_ERR_LEAVES_NOT_POW2, nullptr);
    std::vector<uint8_t> digests;
    SMI_TRY(leaf_digests(ctx, last.data(), n_last, digests));
    uint8_t last_root[32];
    SMI_TRY(smi_merkle_commit(ctx, digests.data(), n_last, last_root));
    if (memcmp(last_root, roots.back(), 32) != 0) return reject(ctx, accept, "last codeword is not well formed");
    const size_t degree_bound = n_last / cfg.expansion_factor;                      // :360-365
    if (degree_bound == 0) return reject(ctx, accept, "last codeword too small");
    uint64_t last_omega = cfg.omega % p, last_offset = cfg.offset % p;
    for (uint64_t i = 0; i + 1 < R; i++) {
        last_omega = mulm(last_omega, last_omega, p);
        last_offset = mulm(last_offset, last_offset, p);
    }
    // The reference interpolates over the point list offset_L * omega_L^i; the transform needs that list
    // to be the coset of the 2^k-th roots (any prover that folded a codeword over a proper domain has it).
    if (n_last > ((uint64_t)1 << ctx->fs.K) || last_omega != h_root(ctx, ilog2(n_last)) || last_offset == 0)
        return smi_fail(ctx, SMI_ERR_NOT_GEOMETRIC, "Fri::verify: the last layer's domain is not offset * <primitive root>");
    for (size_t i = 0; i < n_last; i++)
        if (last[i] >= p) return reject(ctx, accept, "re-evaluated codeword does not match original!");   // :384-390 on an unreduced value
    std::vector<uint64_t> coeffs(n_last), re_eval(n_last);
    if (n_last > 1) {
        SMI_TRY(smi_intt(ctx, last.data(), coeffs.data(), ilog2(n_last), last_offset));
        SMI_TRY(smi_coset_ntt(ctx, coeffs.data(), n_last, re_eval.data(), ilog2(n_last), last_offset));
        if (re_eval != last) return reject(ctx, accept, "re-evaluated codeword does not match original!");
    } else {
        coeffs = last;
    }
    for (size_t i = degree_bound; i < n_last; i++)                                  // :392-397: degree <= degree_bound - 1
        if (coeffs[i] != 0) return reject(ctx, accept, "last codeword does not correspond to polynomial of low enough degree");

    // index sampling (:400-405, :168-213), counters hashed a batch at a time
    const uint64_t size = N >> 1, reduced_size = N >> (R - 1);
    if (t > 2 * reduced_size) return smi_fail(ctx, SMI_ERR_SAMPLE_ENTROPY, nullptr);
    if (t > reduced_size) return smi_fail(ctx, SMI_ERR_SAMPLE_TOO_MANY, nullptr);
    uint64_t seed_ch = 0;
    SMI_TRY(challenge_of(ctx, transcript, &seed_ch));
    uint8_t seed[32], seed_msg[8];
    for (int k = 0; k < 8; k++) seed_msg[k] = (uint8_t)(seed_ch >> (8 * k));
    SMI_TRY(smi_hash_bytes(ctx, seed_msg, 8, seed));                                // Hash::from_u64
    std::vector<uint64_t> top, reduced;
    for (uint32_t counter = 0; top.size() < t;) {
        const size_t run = 2 * (size_t)(t - top.size()) + 8;
        std::vector<uint8_t> msgs(36 * run), dig(32 * run);
        for (size_t k = 0; k < run; k++) {
            memcpy(&msgs[36 * k], seed, 32);
            for (int b = 0; b < 4; b++) msgs[36 * k + 32 + b] = (uint8_t)((counter + k) >> (8 * b));
        }
        SMI_TRY(smi_hash_bytes_batch(ctx, msgs.data(), run, 36, dig.data()));
        for (size_t k = 0; k < run && top.size() < t; k++, counter++) {
            uint64_t acc = 0;                                                       // sample_index: the last eight digest bytes, big-endian
            for (int b = 24; b < 32; b++) acc = (acc << 8) | dig[32 * k + b];
            const uint64_t index = acc % size, ri = index % reduced_size;
            bool seen = false;
            for (uint64_t q : reduced) seen |= q == ri;
            if (!seen) {
                top.push_back(index);
                reduced.push_back(ri);
            }
        }
    }
    if (top_out) *top_out = top;

    uint64_t om = cfg.omega % p, off = cfg.offset % p;
    for (uint64_t r = 0; r + 1 < R; r++) {                                          // :408-502
        const uint64_t half = N >> (r + 1);
        std::vector<uint64_t> ci(t), bi(t), aa(t), bb(t), cc(t);
        for (uint64_t s = 0; s < t; s++) {
            ci[s] = top[s] % half;
            bi[s] = ci[s] + half;
            const Obj *o = pop();
            if (!o || o->tag != 2) return reject(ctx, accept, "Failed to extract triple values");
            if (o->count != 3) return reject(ctx, accept, "Expected triple of values");
            aa[s] = get_u64(o->p);
            bb[s] = get_u64(o->p + 8);
            cc[s] = get_u64(o->p + 16);
            if (r == 0) {
                if (pv_idx && pv_val) {
                    pv_idx->push_back(ci[s]); pv_val->push_back(aa[s]);
                    pv_idx->push_back(bi[s]); pv_val->push_back(bb[s]);
                }
                if (layer0_ab) {
                    layer0_ab->push_back(aa[s]);
                    layer0_ab->push_back(bb[s]);
                }
            }
            // test_colinearity (:507-525): (y1 - y0)(x2 - x0) == (y2 - y0)(x1 - x0)
            const uint64_t ax = mulm(off, powm(om, ci[s], p), p), bx = mulm(off, powm(om, bi[s], p), p), cx = alphas[r] % p;
            if (mulm(subm(bb[s], aa[s], p), subm(cx, ax, p), p) != mulm(subm(cc[s], aa[s], p), subm(bx, ax, p), p))
                return reject(ctx, accept, "colinearity check failure");
        }
        // The 3t authentication paths.  The reference pops and verifies them one at a time in the order
        // (test 0: a, b, c), (test 1: a, b, c), ... and stops at the first failure of either kind; here the paths
        // are verified in one device batch per (a, b, c), so the pops run first, up to the first one that fails,
        // and the verdict is the earliest failure in the reference's order.
        static const char *const miss[3] = {"Failed to extract path for aa", "Failed to extract path for bb", "Failed to extract path for cc"};
        static const char *const bad[3] = {"merkle authentication path verification fails for aa", "merkle authentication path verification fails for bb",
                                           "merkle authentication path verification fails for cc"};
        const uint32_t want_depth[3] = {ilog2(2 * half), ilog2(2 * half), ilog2(half)};
        std::vector<std::vector<uint8_t>> paths(3);
        std::vector<uint64_t> have(3, 0);        // how many paths of each kind were popped
        uint64_t stop_at = 3 * t;                // position (3 s + w) of the first pop that failed
        const char *stop_why = nullptr;
        for (uint64_t s = 0; s < t && !stop_why; s++)
            for (int w = 0; w < 3; w++) {
                const Obj *o = pop();
                if (!o || o->tag != 3) { stop_at = 3 * s + w; stop_why = miss[w]; break; }
                if (o->count != want_depth[w]) { stop_at = 3 * s + w; stop_why = bad[w]; break; }   // wrong length: the recomputed root cannot match
                paths[w].insert(paths[w].end(), o->p, o->p + 32 * o->count);
                have[w]++;
            }
        const std::vector<uint64_t> *vals[3] = {&aa, &bb, &cc}, *idxs[3] = {&ci, &bi, &ci};
        const uint8_t *rt[3] = {roots[r], roots[r], roots[r + 1]};
        uint64_t first_bad = stop_at;
        const char *why = stop_why;
        for (int w = 0; w < 3; w++) {
            const uint64_t k = have[w];
            if (!k) continue;
            std::vector<uint8_t> leaf, ok(k);
            SMI_TRY(leaf_digests(ctx, vals[w]->data(), k, leaf));
            if (want_depth[w])
                SMI_TRY(smi_merkle_verify_batch(ctx, leaf.data(), idxs[w]->data(), paths[w].data(), k, want_depth[w], rt[w], ok.data()));
            else
                for (uint64_t s = 0; s < k; s++) ok[s] = memcmp(&leaf[32 * s], rt[w], 32) == 0;   // a one-leaf tree: the leaf is the root
            for (uint64_t s = 0; s < k; s++)
                if (!ok[s] && 3 * s + w < first_bad) {
                    first_bad = 3 * s + w;
                    why = bad[w];
                }
        }
        if (why) return reject(ctx, accept, why);
        om = mulm(om, om, p);
        off = mulm(off, off, p);
    }
    *accept = 1;
    if (used) *used = at;
    return SMI_OK;
}
size_t fri_object_count(const smi_fri_cfg &cfg) {
    uint64_t R = 0;
    smi_fri_num_rounds(&cfg, &R);
    return (size_t)(R + 1 + (R ? R - 1 : 0) * 4 * cfg.num_colinearity_tests);
}
}  // namespace

int smi_fri_verify(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint8_t *proof, size_t proof_len, int *accept, uint64_t *pv_indices,
                   uint64_t *pv_values, size_t *n_pv) {
    if (!ctx || !cfg || (!proof && proof_len) || !accept) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    *accept = 0;
    if (n_pv) *n_pv = 0;
    SMI_TRY(smi_fri_check(ctx, cfg));
    size_t end = 0;
    const std::vector<Obj> objs = parse(proof, proof_len, (size_t)-1, &end);
    std::vector<uint64_t> pi, pv;
    const int rc = fri_verify_objs(ctx, *cfg, objs, accept, nullptr, &pi, &pv, nullptr, nullptr);
    if (n_pv) *n_pv = pi.size();       // like the reference's &mut Vec: what was pushed before a rejection stays
    if (pv_indices && !pi.empty()) memcpy(pv_indices, pi.data(), 8 * pi.size());
    if (pv_values && !pv.empty()) memcpy(pv_values, pv.data(), 8 * pv.size());
    return rc;
}

int smi_stark_verify(smi_ctx *ctx, const smi_stark_cfg *cfg, const uint8_t *column_roots, const uint8_t *proof, size_t proof_len,
                     int *accept) {
    if (!ctx || !cfg || !column_roots || (!proof && proof_len) || !accept) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    *accept = 0;
    const uint32_t W = cfg->n_cols, logN = cfg->log_n + cfg->log_blowup;
    if (!W || W > 64 || cfg->row_leaves) return smi_fail(ctx, SMI_ERR_BAD_ARG, "stark_verify: 1..64 column trees");
    // Without the column openings the proof is Fri::prove's bytes and nothing else: no relation to column_roots
    // could be checked, so nothing is "verified" here (include/stark_mi.h).
    if (!cfg->open_columns)
        return smi_fail(ctx, SMI_ERR_COLUMNS_NOT_BOUND, "stark_verify: proof made without open_columns; use smi_fri_verify for the FRI part");
    if (cfg->log_blowup < 2) return smi_fail(ctx, SMI_ERR_EXPANSION_TOO_SMALL, nullptr);
    if (logN > ctx->fs.K) return smi_fail(ctx, ctx->fs.F.p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "LDE domain too large");
    const uint64_t p = ctx->fs.F.p, N = 1ull << logN, t = cfg->num_colinearity_tests;
    smi_fri_cfg fc;
    fc.omega = h_root(ctx, logN);
    fc.offset = cfg->lde_offset;
    fc.domain_length = N;
    fc.expansion_factor = 1ull << cfg->log_blowup;
    fc.num_colinearity_tests = t;
    size_t end = 0;
    const std::vector<Obj> objs = parse(proof, proof_len, fri_object_count(fc), &end);
    std::vector<uint64_t> top, ab;
    size_t used = 0;
    SMI_TRY(fri_verify_objs(ctx, fc, objs, accept, &top, nullptr, nullptr, &ab, &used));
    if (!*accept) return SMI_OK;
    // ---- the column openings (mgpu_core.h layout): rows, then paths
    *accept = 0;
    const size_t rec = 9 + 8 * (size_t)W, prec = 9 + 32 * (size_t)logN, need = t * 2 * rec + t * W * 2 * prec;
    if (proof_len - end != need) return reject(ctx, accept, "column openings: wrong length");
    const uint8_t *ext = proof + end, *pathsb = ext + t * 2 * rec;
    // weight c = FiatShamir::challenge after absorbing roots[0..c] (fresh transcript)
    std::vector<uint64_t> weights(W);
    std::vector<uint8_t> transcript;
    for (uint32_t c = 0; c < W; c++) {
        transcript.insert(transcript.end(), column_roots + 32 * c, column_roots + 32 * c + 32);
        SMI_TRY(challenge_of(ctx, transcript, &weights[c]));
    }
    std::vector<uint64_t> rows(t * 2 * W);
    for (uint64_t s = 0; s < t; s++)
        for (int k = 0; k < 2; k++) {
            const uint8_t *r = ext + (2 * s + k) * rec;
            if (r[0] != 2 || get_u64(r + 1) != W) return reject(ctx, accept, "column openings: malformed row");
            uint64_t acc = 0;
            for (uint32_t c = 0; c < W; c++) {
                const uint64_t v = get_u64(r + 9 + 8 * c);
                rows[(2 * s + k) * W + c] = v;
                acc = (acc + mulm(weights[c] % p, v % p, p)) % p;
            }
            if (acc != ab[2 * s + k] % p) return reject(ctx, accept, "column openings: the weighted sum is not the codeword value");
        }
    const uint64_t half = N / 2;
    for (uint32_t c = 0; c < W; c++) {
        std::vector<uint64_t> vals(2 * t), idx(2 * t);
        std::vector<uint8_t> paths(2 * t * 32 * (size_t)logN), leaf, ok(2 * t ? 2 * t : 1);
        for (uint64_t s = 0; s < t; s++)
            for (int k = 0; k < 2; k++) {
                const uint8_t *pr = pathsb + ((s * W + c) * 2 + k) * prec;
                if (pr[0] != 3 || get_u64(pr + 1) != logN) return reject(ctx, accept, "column openings: malformed path");
                vals[2 * s + k] = rows[(2 * s + k) * W + c];
                idx[2 * s + k] = top[s] % half + (k ? half : 0);
                memcpy(&paths[(2 * s + k) * 32 * (size_t)logN], pr + 9, 32 * (size_t)logN);
            }
        SMI_TRY(leaf_digests(ctx, vals.data(), 2 * t, leaf));
        if (t) SMI_TRY(smi_merkle_verify_batch(ctx, leaf.data(), idx.data(), paths.data(), 2 * t, logN, column_roots + 32 * c, ok.data()));
        for (uint64_t s = 0; s < 2 * t; s++)
            if (!ok[s]) return reject(ctx, accept, "column openings: authentication path does not verify");
    }
    *accept = 1;
    return SMI_OK;
}
