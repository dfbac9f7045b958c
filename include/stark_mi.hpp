// stark_mi.hpp -- header-only C++17 host mirror of the reference's types over the C ABI.
//
// The reference is Rust (0xSooki/stark-rs) and the build image has no Rust toolchain, so the host
// side above the C ABI is written in C++: the same names, argument meaning and error behaviour as
// the reference's `FiniteField`, `FieldElement`, `Polynomial`, `Hash`, `MerkleTree`, `FiatShamir`,
// `ProofStream`, `Fri`, `Trace` for the hot path, with every heavy method a call into
// libstarkmi.so (include/stark_mi.h).  A reference panic becomes starkmi::Panic whose what() is
// the reference's message.  tests/cpp/fri_mirror_test.cpp replays src/fri.rs:533-693 on it.
#pragma once
#include <cstring>
#include <memory>
#include <array>
#include <stdexcept>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "stark_mi.h"

namespace starkmi {

struct Panic : std::runtime_error {
    int status;
    Panic(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

inline void check(int status, const smi_ctx *ctx = nullptr) {
    if (status == SMI_OK) return;
    std::string msg = smi_status_string(status);
    if (ctx && status <= SMI_ERR_BAD_ARG) msg += std::string(": ") + smi_last_error(ctx);
    throw Panic(status, msg);
}

// One GPU context per modulus, created on first use (single-owner, like the Rust binding's thread_local).
inline smi_ctx *context(uint64_t p) {
    static smi_ctx *ref = nullptr, *second = nullptr;
    smi_ctx **slot = p == 998244353ull ? &ref : &second;
    if (!*slot) check(smi_ctx_create(p, 3, 0, slot));
    return *slot;
}

struct FieldElement;

// src/ff.rs:9-12, 108-233
struct FiniteField {
    uint64_t p;
    explicit FiniteField(uint64_t p_ = 998244353ull) : p(p_) {}
    bool operator==(const FiniteField &o) const { return p == o.p; }
    uint64_t modulus() const { return p; }
    inline FieldElement new_element(uint64_t v) const;
    inline FieldElement zero() const;
    inline FieldElement one() const;
    inline FieldElement add(const FieldElement &l, const FieldElement &r) const;
    inline FieldElement sub(const FieldElement &l, const FieldElement &r) const;
    inline FieldElement mul(const FieldElement &l, const FieldElement &r) const;
    inline FieldElement neg(const FieldElement &x) const;
    inline FieldElement inv(const FieldElement &x) const;
    inline FieldElement div(const FieldElement &l, const FieldElement &r) const;
    inline FieldElement exp(const FieldElement &b, uint64_t e) const;
    inline FieldElement g() const;
    inline FieldElement prim_nth_root(uint64_t n) const;
    smi_ctx *ctx() const { return context(p); }
};

// src/ff.rs:24-28
struct FieldElement {
    uint64_t value;
    FiniteField field;
    bool operator==(const FieldElement &o) const { return value == o.value && field == o.field; }
    bool operator!=(const FieldElement &o) const { return !(*this == o); }
    FieldElement operator+(const FieldElement &o) const { return field.add(*this, o); }
    FieldElement operator-(const FieldElement &o) const { return field.sub(*this, o); }
    FieldElement operator*(const FieldElement &o) const { return field.mul(*this, o); }
    FieldElement operator/(const FieldElement &o) const { return field.div(*this, o); }
    FieldElement pow(uint64_t e) const { return field.exp(*this, e); }
};

inline FieldElement FiniteField::new_element(uint64_t v) const { return FieldElement{v, *this}; }  // unreduced, ff.rs:113-118
inline FieldElement FiniteField::zero() const { return FieldElement{0, *this}; }
inline FieldElement FiniteField::one() const { return FieldElement{1, *this}; }
inline FieldElement FiniteField::add(const FieldElement &l, const FieldElement &r) const {
    return {(uint64_t)(((unsigned __int128)l.value + r.value) % p), *this};
}
inline FieldElement FiniteField::sub(const FieldElement &l, const FieldElement &r) const {
    return {(uint64_t)((((unsigned __int128)p + l.value) - r.value) % p), *this};
}
inline FieldElement FiniteField::mul(const FieldElement &l, const FieldElement &r) const {
    return {(uint64_t)(((unsigned __int128)l.value * r.value) % p), *this};
}
inline FieldElement FiniteField::neg(const FieldElement &x) const { return {(p - x.value) % p, *this}; }
inline FieldElement FiniteField::inv(const FieldElement &x) const {
    uint64_t out = 0;
    check(smi_ff_inv(ctx(), x.value, &out));  // "no inverse", ff.rs:171
    return {out, *this};
}
inline FieldElement FiniteField::div(const FieldElement &l, const FieldElement &r) const {
    if (r.value == 0) throw Panic(SMI_ERR_DIV_BY_ZERO, smi_status_string(SMI_ERR_DIV_BY_ZERO));
    return mul(l, inv(r));
}
inline FieldElement FiniteField::exp(const FieldElement &b, uint64_t e) const {
    uint64_t out = 0;
    check(smi_ff_exp(ctx(), b.value, e, &out));
    return {out, *this};
}
inline FieldElement FiniteField::g() const {
    if (p != 998244353ull) throw Panic(SMI_ERR_WRONG_FIELD, smi_status_string(SMI_ERR_WRONG_FIELD));  // ff.rs:192
    return {3, *this};
}
inline FieldElement FiniteField::prim_nth_root(uint64_t n) const {
    if (p != 998244353ull) throw Panic(SMI_ERR_WRONG_FIELD, smi_status_string(SMI_ERR_WRONG_FIELD));  // ff.rs:216
    uint64_t out = 0;
    check(smi_prim_nth_root(ctx(), n, &out));
    return {out, *this};
}

inline std::vector<uint64_t> values_of(const std::vector<FieldElement> &v) {
    std::vector<uint64_t> out(v.size());
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i].value;
    return out;
}
inline std::vector<FieldElement> elements_of(const std::vector<uint64_t> &v, const FiniteField &f) {
    std::vector<FieldElement> out;
    out.reserve(v.size());
    for (uint64_t x : v) out.push_back(f.new_element(x));
    return out;
}
inline uint32_t log2_exact(size_t n) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

// src/univariate/mod.rs:8-153, interpolate.rs, eval.rs
struct Polynomial {
    std::vector<FieldElement> coeffs;
    FiniteField field;
    Polynomial(std::vector<FieldElement> c, FiniteField f) : coeffs(std::move(c)), field(f) {}
    long deg() const {
        long d = -1;
        for (size_t i = 0; i < coeffs.size(); i++)
            if (coeffs[i].value != 0) d = (long)i;
        return d;
    }
    bool is_zero() const { return deg() == -1; }
    bool operator==(const Polynomial &o) const {  // mod.rs:13-39: trailing zeros ignored
        if (deg() != o.deg()) return false;
        for (long i = 0; i <= deg(); i++)
            if (coeffs[i] != o.coeffs[i]) return false;
        return true;
    }
    FieldElement eval(const FieldElement &x) const {  // eval.rs:6-14 (single point: host)
        FieldElement xi = field.one(), val = field.zero();
        for (const FieldElement &c : coeffs) {
            val = val + c * xi;
            xi = xi * x;
        }
        return val;
    }
    // eval.rs:16-21 on a geometric domain offset*omega^k (the fast-path contract)
    std::vector<FieldElement> eval_domain(const std::vector<FieldElement> &domain) const {
        if (domain.empty()) return {};
        const FiniteField f = domain[0].field;
        std::vector<uint64_t> d = values_of(domain);
        uint64_t offset = 0;
        check(smi_domain_is_geometric(f.ctx(), d.data(), d.size(), &offset));
        const size_t nc = (size_t)(deg() + 1);
        std::vector<uint64_t> c = values_of(coeffs), out(d.size());
        c.resize(nc);
        check(smi_coset_ntt(f.ctx(), c.data(), nc, out.data(), log2_exact(d.size()), offset), f.ctx());
        return elements_of(out, f);
    }
    // interpolate.rs:6-44 on a geometric domain
    static Polynomial interpolate_domain(const std::vector<FieldElement> &domain, const std::vector<FieldElement> &values) {
        if (domain.size() != values.size()) throw Panic(SMI_ERR_LEN_MISMATCH, smi_status_string(SMI_ERR_LEN_MISMATCH));
        if (domain.empty()) throw Panic(SMI_ERR_EMPTY_DOMAIN, smi_status_string(SMI_ERR_EMPTY_DOMAIN));
        const FiniteField f = domain[0].field;
        std::vector<uint64_t> d = values_of(domain), v = values_of(values), c(v.size());
        uint64_t offset = 0;
        check(smi_domain_is_geometric(f.ctx(), d.data(), d.size(), &offset));
        check(smi_intt(f.ctx(), v.data(), c.data(), log2_exact(v.size()), offset), f.ctx());
        bool all_zero = true;
        for (uint64_t x : v) all_zero = all_zero && x == 0;
        if (v.size() > 1 && all_zero) return Polynomial({}, f);
        return Polynomial(elements_of(c, f), f);
    }
    Polynomial scale(const FieldElement &factor) const {  // mod.rs:99-113
        std::vector<uint64_t> c = values_of(coeffs), out(c.size());
        if (!c.empty()) check(smi_poly_scale(field.ctx(), c.data(), c.size(), factor.value % field.p, out.data()), field.ctx());
        return Polynomial(elements_of(out, field), field);
    }
    static Polynomial mul(const Polynomial &l, const Polynomial &r) {  // mul.rs:6-29, NTT product on the device
        std::vector<uint64_t> a = values_of(l.coeffs), b = values_of(r.coeffs), out(a.size() + b.size() + 1);
        size_t n = 0;
        check(smi_poly_mul(l.field.ctx(), a.data(), a.size(), b.data(), b.size(), out.data(), &n), l.field.ctx());
        out.resize(n);
        return Polynomial(elements_of(out, l.field), l.field);
    }
    // div.rs:6-42 -> (quotient, remainder); panics "No division by zero"
    static std::pair<Polynomial, Polynomial> div(const Polynomial &numer, const Polynomial &denom) {
        std::vector<uint64_t> a = values_of(numer.coeffs), b = values_of(denom.coeffs);
        std::vector<uint64_t> q(a.size() + 1), r(std::max(a.size(), b.size()) + 1);
        size_t nq = 0, nr = 0;
        check(smi_poly_div(numer.field.ctx(), a.data(), a.size(), b.data(), b.size(), q.data(), &nq, r.data(), &nr), numer.field.ctx());
        q.resize(nq);
        r.resize(nr);
        return {Polynomial(elements_of(q, numer.field), numer.field), Polynomial(elements_of(r, numer.field), numer.field)};
    }
};

// src/hash.rs:1-46
struct Hash {
    uint8_t b[32];
    bool operator==(const Hash &o) const { return std::memcmp(b, o.b, 32) == 0; }
    bool operator!=(const Hash &o) const { return !(*this == o); }
    static Hash from_bytes(const uint8_t *data, size_t len) {
        Hash h;
        check(smi_hash_bytes(context(998244353ull), data, len, h.b));
        return h;
    }
    static Hash from_field_elements(const std::vector<uint64_t> &e) {
        std::vector<uint8_t> buf(e.size() * 8);
        for (size_t i = 0; i < e.size(); i++)
            for (int k = 0; k < 8; k++) buf[8 * i + k] = (uint8_t)(e[i] >> (8 * k));
        return from_bytes(buf.data(), buf.size());
    }
    static Hash from_u64(uint64_t v) { return from_field_elements({v}); }
    static Hash combine(const Hash &l, const Hash &r) {
        uint8_t in[64];
        std::memcpy(in, l.b, 32);
        std::memcpy(in + 32, r.b, 32);
        Hash h;
        check(smi_hash_combine_pairs(context(998244353ull), in, 1, h.b));
        return h;
    }
};

// src/merkle.rs:4-97; the tree stays on the device
class MerkleTree {
    std::shared_ptr<smi_tree> t_;
    smi_ctx *ctx_;

  public:
    std::vector<Hash> leaves;
    Hash root;
    explicit MerkleTree(const std::vector<Hash> &lv) : ctx_(context(998244353ull)), leaves(lv) {
        smi_tree *t = nullptr;
        check(smi_merkle_new(ctx_, lv.empty() ? nullptr : lv[0].b, lv.size(), &t), ctx_);
        t_.reset(t, smi_merkle_free);
        check(smi_merkle_root(ctx_, t, root.b), ctx_);
    }
    const Hash &get_root() const { return root; }
    static Hash commit(const std::vector<Hash> &lv) {
        Hash h;
        check(smi_merkle_commit(context(998244353ull), lv.empty() ? nullptr : lv[0].b, lv.size(), h.b));
        return h;
    }
    std::vector<Hash> open(size_t index) const {
        std::vector<Hash> path(64);
        size_t depth = 0;
        check(smi_merkle_open(ctx_, t_.get(), index, path[0].b, &depth), ctx_);
        path.resize(depth);
        return path;
    }
    static bool verify(const Hash &leaf, size_t index, const std::vector<Hash> &proof, const Hash &root) {
        Hash cur = leaf;  // merkle.rs:82-96
        for (const Hash &sib : proof) {
            cur = index % 2 == 0 ? Hash::combine(cur, sib) : Hash::combine(sib, cur);
            index /= 2;
        }
        return cur == root;
    }
};

// src/fiat_shamir.rs:4-26
struct FiatShamir {
    std::vector<uint8_t> transcript;
    void absorb(const uint8_t *data, size_t len) { transcript.insert(transcript.end(), data, data + len); }
    FieldElement challenge(const FiniteField &field) const {
        Hash h = Hash::from_bytes(transcript.data(), transcript.size());
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)h.b[i] << (8 * i);
        return field.new_element(v);  // unreduced
    }
};

// src/stream.rs:8-14
struct ProofObject {
    enum Tag { MerkleRoot = 0, FieldElementTag = 1, FieldElements = 2, MerklePath = 3 } tag;
    Hash hash{};
    std::vector<uint64_t> elements;
    std::vector<Hash> path;
};

// src/stream.rs:4-168
struct ProofStream {
    std::vector<ProofObject> objects;
    size_t head = 0;
    void push(ProofObject o) { objects.push_back(std::move(o)); }
    const ProofObject *pop() { return head < objects.size() ? &objects[head++] : nullptr; }
    std::vector<uint8_t> serialize() const {
        std::vector<uint8_t> out;
        auto put64 = [&](uint64_t v) { for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i))); };
        for (const ProofObject &o : objects) {
            out.push_back((uint8_t)o.tag);
            switch (o.tag) {
            case ProofObject::MerkleRoot: out.insert(out.end(), o.hash.b, o.hash.b + 32); break;
            case ProofObject::FieldElementTag: put64(o.elements[0]); break;
            case ProofObject::FieldElements: put64(o.elements.size()); for (uint64_t v : o.elements) put64(v); break;
            case ProofObject::MerklePath: put64(o.path.size()); for (const Hash &h : o.path) out.insert(out.end(), h.b, h.b + 32); break;
            }
        }
        return out;
    }
    static ProofStream deserialize(const std::vector<uint8_t> &b) {
        ProofStream s;
        size_t i = 0, n = b.size();
        auto get64 = [&](size_t at) { uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)b[at + k] << (8 * k); return v; };
        while (i < n) {
            uint8_t tag = b[i++];
            ProofObject o;
            if (tag == 0) {
                if (i + 32 <= n) { o.tag = ProofObject::MerkleRoot; std::memcpy(o.hash.b, &b[i], 32); i += 32; s.push(o); }
            } else if (tag == 1) {
                if (i + 8 <= n) { o.tag = ProofObject::FieldElementTag; o.elements = {get64(i)}; i += 8; s.push(o); }
            } else if (tag == 2) {
                if (i + 8 <= n) {
                    uint64_t len = get64(i); i += 8;
                    o.tag = ProofObject::FieldElements;
                    for (uint64_t k = 0; k < len && i + 8 <= n; k++) { o.elements.push_back(get64(i)); i += 8; }
                    s.push(o);
                }
            } else if (tag == 3) {
                if (i + 8 <= n) {
                    uint64_t len = get64(i); i += 8;
                    o.tag = ProofObject::MerklePath;
                    for (uint64_t k = 0; k < len && i + 32 <= n; k++) { Hash h; std::memcpy(h.b, &b[i], 32); o.path.push_back(h); i += 32; }
                    s.push(o);
                }
            } else break;
        }
        return s;
    }
};

// src/fri.rs:8-504
class Fri {
  public:
    FieldElement offset, omega;
    size_t domain_length;
    FiniteField field;
    size_t expansion_factor, num_colinearity_tests;
    Fri(FieldElement omega_, FieldElement offset_, size_t domain_length_, size_t expansion_factor_, size_t t)
        : offset(offset_), omega(omega_), domain_length(domain_length_), field(omega_.field), expansion_factor(expansion_factor_),
          num_colinearity_tests(t) {
        cfg_ = smi_fri_cfg{omega.value, offset.value, domain_length, expansion_factor, t};
        check(smi_fri_check(field.ctx(), &cfg_));  // asserts of fri.rs:37-45
    }
    uint64_t num_rounds() const {
        uint64_t r = 0;
        check(smi_fri_num_rounds(&cfg_, &r));
        return r;
    }
    std::vector<FieldElement> fold_codeword(const std::vector<FieldElement> &cw, const FieldElement &alpha, const FieldElement &off,
                                            const FieldElement &om) const {
        std::vector<uint64_t> in = values_of(cw), out(cw.size() / 2);
        check(smi_fri_fold(field.ctx(), in.data(), in.size(), alpha.value, off.value, om.value, out.data()), field.ctx());
        return elements_of(out, field);
    }
    // fri.rs:250-311: fills proof_stream, absorbs the roots into fiat_shamir, returns the top-level indices
    std::vector<size_t> prove(const std::vector<FieldElement> &initial_codeword, FiatShamir &fiat_shamir, ProofStream &proof_stream) const {
        if (!fiat_shamir.transcript.empty()) throw Panic(SMI_ERR_BAD_ARG, "bad argument: the device transcript starts empty");
        std::vector<uint64_t> cw = values_of(initial_codeword), top(num_colinearity_tests + 1);
        uint8_t *proof = nullptr;
        size_t len = 0;
        check(smi_fri_prove(field.ctx(), &cfg_, cw.data(), cw.size(), &proof, &len, top.data()), field.ctx());
        std::vector<uint8_t> bytes(proof, proof + len);
        smi_free(proof);
        for (const ProofObject &o : ProofStream::deserialize(bytes).objects) {
            if (o.tag == ProofObject::MerkleRoot) fiat_shamir.absorb(o.hash.b, 32);
            proof_stream.push(o);
        }
        return std::vector<size_t>(top.begin(), top.begin() + num_colinearity_tests);
    }

    // Fri::sample_indices (src/fri.rs:176-213); digests from the device hash
    std::vector<size_t> sample_indices(const Hash &seed, size_t size, size_t reduced_size, size_t number) const {
        if (number > 2 * reduced_size) throw Panic(SMI_ERR_SAMPLE_ENTROPY, "not enough entropy in indices wrt last codeword");
        if (number > reduced_size) throw Panic(SMI_ERR_SAMPLE_TOO_MANY, "cannot sample more indices than available in last codeword");
        std::vector<size_t> indices, reduced;
        for (uint32_t counter = 0; indices.size() < number; counter++) {
            uint8_t msg[36];
            std::memcpy(msg, seed.b, 32);
            for (int k = 0; k < 4; k++) msg[32 + k] = (uint8_t)(counter >> (8 * k));
            const Hash h = Hash::from_bytes(msg, 36);
            uint64_t acc = 0;   // sample_index (:168-174): the last eight digest bytes, big-endian
            for (int k = 24; k < 32; k++) acc = (acc << 8) | h.b[k];
            const size_t index = (size_t)(acc % size), r = index % reduced_size;
            if (std::find(reduced.begin(), reduced.end(), r) == reduced.end()) {
                indices.push_back(index);
                reduced.push_back(r);
            }
        }
        return indices;
    }
    // Fri::verify (src/fri.rs:313-504): the reference's control flow on the host; leaf hashes and
    // Merkle paths checked in batches on the device (smi_hash_leaves, smi_merkle_verify_batch) and
    // the last-layer low-degree test as an inverse NTT instead of the reference's O(L^3) Lagrange
    // interpolation.  false where the reference prints and returns false.
    bool verify(ProofStream &proof_stream, FiatShamir &fiat_shamir, std::vector<std::pair<size_t, FieldElement>> &polynomial_values) const {
        smi_ctx *ctx = field.ctx();
        const uint64_t p = field.p, t = num_colinearity_tests, R = num_rounds();
        auto mulm = [&](uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % p); };
        auto powm = [&](uint64_t b, uint64_t e) { uint64_t r = 1 % p; b %= p; while (e) { if (e & 1) r = mulm(r, b); b = mulm(b, b); e >>= 1; } return r; };
        std::vector<Hash> roots;
        std::vector<uint64_t> alphas;
        for (uint64_t i = 0; i < R; i++) {   // :325-334
            const ProofObject *o = proof_stream.pop();
            if (!o || o->tag != ProofObject::MerkleRoot) return false;
            roots.push_back(o->hash);
            fiat_shamir.absorb(o->hash.b, 32);
            alphas.push_back(fiat_shamir.challenge(field).value);
        }
        const ProofObject *lo = proof_stream.pop();   // :337-342
        if (!lo || lo->tag != ProofObject::FieldElements || roots.empty()) return false;
        const std::vector<uint64_t> last = lo->elements;
        const size_t n_last = last.size();
        if (n_last == 0 || (n_last & (n_last - 1))) throw Panic(SMI_ERR_LEAVES_NOT_POW2, "Number of leaves must be power of 2");
        std::vector<uint8_t> digests(32 * n_last);
        Hash last_root;
        check(smi_hash_leaves(ctx, last.data(), n_last, digests.data()), ctx);   // :349-357
        check(smi_merkle_commit(ctx, digests.data(), n_last, last_root.b), ctx);
        if (last_root != roots.back()) return false;
        const size_t degree_bound = n_last / expansion_factor;   // :360-365
        if (degree_bound == 0) return false;
        uint64_t last_omega = omega.value, last_offset = offset.value;
        for (uint64_t i = 0; i + 1 < R; i++) { last_omega = mulm(last_omega, last_omega); last_offset = mulm(last_offset, last_offset); }
        uint64_t want_root = 0;
        check(smi_prim_nth_root(ctx, n_last, &want_root), ctx);
        if (last_omega != want_root) return false;   // the domain of the last layer must be offset * <omega_L>
        std::vector<uint64_t> coeffs(n_last), re_eval(n_last);
        if (n_last > 1) {
            check(smi_intt(ctx, last.data(), coeffs.data(), log2_exact(n_last), last_offset), ctx);
            check(smi_coset_ntt(ctx, coeffs.data(), n_last, re_eval.data(), log2_exact(n_last), last_offset), ctx);
            if (re_eval != last) return false;       // :384-390
        } else {
            coeffs = last;
        }
        for (size_t i = degree_bound; i < n_last; i++)
            if (coeffs[i] != 0) return false;        // :392-397: degree <= degree_bound - 1
        const Hash seed = Hash::from_u64(fiat_shamir.challenge(field).value);   // :400-405
        const std::vector<size_t> top = sample_indices(seed, domain_length >> 1, domain_length >> (R - 1), t);
        uint64_t om = omega.value, off = offset.value;
        for (uint64_t r = 0; r + 1 < R; r++) {       // :408-502
            const size_t half = domain_length >> (r + 1);
            std::vector<uint64_t> c_idx(t), b_idx(t), aa(t), bb(t), cc(t);
            for (uint64_t s = 0; s < t; s++) {
                c_idx[s] = top[s] % half;
                b_idx[s] = c_idx[s] + half;
                const ProofObject *o = proof_stream.pop();
                if (!o || o->tag != ProofObject::FieldElements || o->elements.size() != 3) return false;
                aa[s] = o->elements[0]; bb[s] = o->elements[1]; cc[s] = o->elements[2];
                if (r == 0) {
                    polynomial_values.emplace_back((size_t)c_idx[s], field.new_element(aa[s]));
                    polynomial_values.emplace_back((size_t)b_idx[s], field.new_element(bb[s]));
                }
                const uint64_t ax = mulm(off, powm(om, c_idx[s])), bx = mulm(off, powm(om, b_idx[s])), cx = alphas[r] % p;
                // test_colinearity (:507-525): (y1-y0)(x2-x0) == (y2-y0)(x1-x0)
                const uint64_t l = mulm((p + bb[s] - aa[s]) % p, (p + cx - ax) % p), rr = mulm((p + cc[s] - aa[s]) % p, (p + bx - ax) % p);
                if (l != rr) return false;
            }
            std::vector<std::vector<uint8_t>> paths(3);   // a, b, c paths, each t x depth x 32
            size_t depth[3] = {0, 0, 0};
            for (uint64_t s = 0; s < t; s++)
                for (int w = 0; w < 3; w++) {
                    const ProofObject *o = proof_stream.pop();
                    if (!o || o->tag != ProofObject::MerklePath) return false;
                    if (s == 0) depth[w] = o->path.size();
                    if (o->path.size() != depth[w]) return false;
                    for (const Hash &h : o->path) paths[w].insert(paths[w].end(), h.b, h.b + 32);
                }
            const std::vector<uint64_t> *vals[3] = {&aa, &bb, &cc}, *idxs[3] = {&c_idx, &b_idx, &c_idx};
            const Hash *rt[3] = {&roots[r], &roots[r], &roots[r + 1]};
            for (int w = 0; w < 3; w++) {
                std::vector<uint8_t> leaf(32 * t), ok(t);
                check(smi_hash_leaves(ctx, vals[w]->data(), t, leaf.data()), ctx);
                check(smi_merkle_verify_batch(ctx, leaf.data(), idxs[w]->data(), paths[w].data(), t, depth[w], rt[w]->b, ok.data()), ctx);
                for (uint8_t f : ok) if (!f) return false;
            }
            om = mulm(om, om);
            off = mulm(off, off);
        }
        return true;
    }

  private:
    smi_fri_cfg cfg_;
};

// src/trace.rs:3-50
struct Trace {
    std::vector<std::vector<__int128>> trace;
    size_t num_columns;
    explicit Trace(const std::vector<std::vector<__int128>> &t) : trace(t), num_columns(t[0].size()) {}
    std::vector<__int128> get_col(size_t j) const {
        std::vector<__int128> c;
        for (const auto &r : trace) c.push_back(r[j]);
        return c;
    }
    static Trace fibonacci(size_t length) {
        std::vector<std::vector<__int128>> rows;
        __int128 a = 1, b = 1;
        for (size_t i = 0; i < length; i++) {
            rows.push_back({a});
            __int128 next = a + b;
            a = b;
            b = next;
        }
        return Trace(rows);
    }
    // build-defined: column-major low-degree extension on the device
    std::vector<std::vector<uint64_t>> lde(const FiniteField &f, uint32_t log_blowup, uint64_t lde_offset) const {
        const size_t n = trace.size();
        std::vector<__int128> rows;   // row-major i128, the layout of Trace.trace (trace.rs:4-7)
        for (const auto &r : trace) rows.insert(rows.end(), r.begin(), r.end());
        std::vector<uint64_t> cols(num_columns * n), out(num_columns * (n << log_blowup));
        check(smi_trace_pack(f.ctx(), rows.data(), n, num_columns, cols.data()));
        check(smi_lde(f.ctx(), cols.data(), (uint32_t)num_columns, log2_exact(n), log_blowup, 1, lde_offset, out.data()), f.ctx());
        std::vector<std::vector<uint64_t>> res(num_columns);
        for (size_t c = 0; c < num_columns; c++) res[c].assign(out.begin() + c * (n << log_blowup), out.begin() + (c + 1) * (n << log_blowup));
        return res;
    }
};

// Multi-GPU prover (one process per GPU; csrc/mgpu.hip): the communicator over the ranks of a node and
// Fri::prove / the build-defined prove / one transform sharded across them.  Device pointers in,
// the reference's serialized ProofStream out -- on every rank.  `id` = smi_mgpu_unique_id's 128 bytes
// from rank 0, carried to the other ranks by the launcher.
class MultiGpu {
  public:
    static std::array<uint8_t, SMI_MGPU_ID_BYTES> unique_id() {
        std::array<uint8_t, SMI_MGPU_ID_BYTES> id{};
        check(smi_mgpu_unique_id(id.data()));
        return id;
    }
    MultiGpu(const FiniteField &f, const std::array<uint8_t, SMI_MGPU_ID_BYTES> &id, int rank, int world) : ctx_(f.ctx()) {
        check(smi_mgpu_create(ctx_, id.data(), rank, world, &m_), ctx_);
    }
    MultiGpu(const MultiGpu &) = delete;
    MultiGpu &operator=(const MultiGpu &) = delete;
    ~MultiGpu() { smi_mgpu_destroy(m_); }
    // Fri::prove (src/fri.rs:250-311) of the codeword this rank holds block `rank` of
    std::vector<size_t> fri_prove(const smi_fri_cfg &cfg, const uint32_t *d_block, size_t block_len, ProofStream &proof_stream) {
        uint8_t *bytes = nullptr;
        size_t len = 0;
        std::vector<uint64_t> top(cfg.num_colinearity_tests ? cfg.num_colinearity_tests : 1);
        check(smi_mgpu_fri_prove(m_, &cfg, d_block, block_len, &bytes, &len, top.data()), ctx_);
        ProofStream got = ProofStream::deserialize(std::vector<uint8_t>(bytes, bytes + len));
        smi_free(bytes);
        for (auto &o : got.objects) proof_stream.push(o);
        return std::vector<size_t>(top.begin(), top.begin() + cfg.num_colinearity_tests);
    }
    // trace -> (column roots, proof bytes): smi_dev_stark_prove over the ranks
    std::vector<uint8_t> stark_prove(const smi_stark_cfg &cfg, const uint32_t *d_trace_cols, std::vector<Hash> &column_roots,
                                     std::vector<size_t> &top_indices) {
        std::vector<uint8_t> roots(32 * (size_t)cfg.n_cols);
        std::vector<uint64_t> top(cfg.num_colinearity_tests ? cfg.num_colinearity_tests : 1);
        uint8_t *bytes = nullptr;
        size_t len = 0;
        check(smi_mgpu_stark_prove(m_, &cfg, d_trace_cols, roots.data(), &bytes, &len, top.data()), ctx_);
        std::vector<uint8_t> proof(bytes, bytes + len);
        smi_free(bytes);
        column_roots.clear();
        for (uint32_t c = 0; c < cfg.n_cols; c++) {
            Hash h;
            std::memcpy(h.b, roots.data() + 32 * c, 32);
            column_roots.push_back(h);
        }
        top_indices.assign(top.begin(), top.begin() + cfg.num_colinearity_tests);
        return proof;
    }
    // one 2^log_n-point transform over the ranks (layouts: include/stark_mi.h, smi_mgpu_ntt)
    void ntt(uint32_t *d_strip, uint32_t *d_out, uint32_t log_n, bool inverse = false, uint64_t offset = 1) {
        check(smi_mgpu_ntt(m_, d_strip, d_out, log_n, inverse ? 1 : 0, offset), ctx_);
    }

  private:
    smi_ctx *ctx_;
    smi_mgpu *m_ = nullptr;
};

}  // namespace starkmi
