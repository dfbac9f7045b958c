// fri_mirror_test.cpp -- the reference's own tests for the hot path, replayed in C++ on the
// host mirror (include/stark_mi.hpp) whose heavy methods run on the GPU through the C ABI.
// Follows src/fri.rs:533-693 (prove -> serialize -> deserialize -> verify, four parameter sets),
// src/merkle.rs:104-133, src/univariate/interpolate.rs:61-163 / eval.rs:101-117 (on geometric
// domains) and mod.rs:439-456.  Fri::verify is the reference's CPU code; here the oracle's
// restatement of it (oracle/stark_oracle.h, test infrastructure) plays that role.
// Built and run by tests/test_gpu_cpp_mirror.py (needs a GPU).
#include <cstdio>
#include <cstdlib>

#include "../../include/stark_mi.hpp"
#include "../../oracle/stark_oracle.h"

using namespace starkmi;

static int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

template <class F> static bool panics_with(F f, const char *needle) {
    try { f(); } catch (const Panic &e) { return std::string(e.what()).find(needle) != std::string::npos; }
    return false;
}

static void fri_case(size_t n, size_t expansion, size_t t, uint64_t offset_v, std::vector<uint64_t> coeffs_v) {
    FiniteField field(998244353);
    FieldElement omega = field.prim_nth_root(n), offset = field.new_element(offset_v);
    Fri fri(omega, offset, n, expansion, t);
    Polynomial poly(elements_of(coeffs_v, field), field);
    std::vector<FieldElement> domain;
    for (size_t i = 0; i < n; i++) domain.push_back(field.mul(offset, field.exp(omega, i)));
    std::vector<FieldElement> codeword = poly.eval_domain(domain);
    for (size_t i = 0; i < n; i += 7) EXPECT(codeword[i] == poly.eval(domain[i]));   // eval_domain order, eval.rs:16-21

    ProofStream proof_stream;
    FiatShamir prover_fiat_shamir;
    std::vector<size_t> top = fri.prove(codeword, prover_fiat_shamir, proof_stream);
    EXPECT(top.size() == t);
    std::vector<uint8_t> proof_bytes = proof_stream.serialize();
    ProofStream verifier_stream = ProofStream::deserialize(proof_bytes);
    EXPECT(verifier_stream.serialize() == proof_bytes);
    EXPECT(prover_fiat_shamir.transcript.size() == 32 * fri.num_rounds());

    // the mirror's own Fri::verify (device hashes / paths / iNTT) accepts, and rejects a tampered proof
    {
        FiatShamir vfs;
        std::vector<std::pair<size_t, FieldElement>> pv;
        EXPECT(fri.verify(verifier_stream, vfs, pv));
        EXPECT(pv.size() == 2 * t);
        for (const auto &iv : pv) EXPECT(iv.second == codeword[iv.first]);   // the opened top-layer values (fri.rs:437-442)
        std::vector<uint8_t> bad = proof_bytes;
        bad[bad.size() / 2] ^= 1;
        ProofStream tampered = ProofStream::deserialize(bad);
        FiatShamir vfs2;
        std::vector<std::pair<size_t, FieldElement>> pv2;
        EXPECT(!fri.verify(tampered, vfs2, pv2));
    }

    so_fri_cfg cfg{field.p, omega.value, offset.value, n, expansion, t};
    int ok = so_fri_verify(&cfg, proof_bytes.data(), proof_bytes.size(), nullptr, nullptr, nullptr);
    if (ok != 1) std::printf("verify rejected: %s %s\n", so_fri_last_reject(), so_last_panic());
    EXPECT(ok == 1);
    std::printf("fri n=%zu expansion=%zu t=%zu rounds=%llu proof=%zu B: %s\n", n, expansion, t,
                (unsigned long long)fri.num_rounds(), proof_bytes.size(), ok == 1 ? "verified" : "REJECTED");
}

int main() {
    const uint64_t P = 998244353;
    FiniteField field(P);

    // src/univariate/div.rs:83-123, :169-178 and mul.rs on the device
    {
        auto P3 = [&](std::vector<uint64_t> v) { return Polynomial(elements_of(v, field), field); };
        auto qr = Polynomial::div(P3({2, 3, 1}), P3({1, 1}));                       // test_division_basic
        EXPECT(qr.first.deg() == 1 && qr.first.coeffs[0].value == 2 && qr.first.coeffs[1].value == 1 && qr.second.is_zero());
        qr = Polynomial::div(P3({1, 0, 1}), P3({1, 1}));                            // test_division_with_remainder
        EXPECT(qr.first.deg() == 1 && qr.second.deg() == 0 && qr.second.coeffs[0].value == 2);
        EXPECT(panics_with([&] { Polynomial::div(P3({1, 2}), P3({})); }, "No division by zero"));
        Polynomial prod = Polynomial::mul(P3({1, 0, 2}), P3({3, 0, 4}));            // test_mul_sparse
        EXPECT(prod == P3({3, 0, 10, 0, 8}));
        qr = Polynomial::div(prod, P3({3, 0, 4}));
        EXPECT(qr.first == P3({1, 0, 2}) && qr.second.is_zero());
    }

    // src/ff.rs panics
    EXPECT(panics_with([&] { field.inv(field.zero()); }, "no inverse"));
    EXPECT(panics_with([&] { field.prim_nth_root(6); }, "n must be a power of two"));
    EXPECT(panics_with([&] { field.prim_nth_root(1ull << 24); }, "n > 2^23 not supported"));
    EXPECT(field.exp(field.new_element(2), 10).value == 1024);

    // interpolate / eval / scale on a coset of the 4th roots (interpolate.rs KAT polynomials)
    {
        FieldElement w = field.prim_nth_root(4), off = field.new_element(3);
        std::vector<FieldElement> domain;
        for (int k = 0; k < 4; k++) domain.push_back(off * field.exp(w, k));
        for (std::vector<uint64_t> c : {std::vector<uint64_t>{0, 0, 1}, {3, 2}, {1, 0, 1}, {P - 2, 5, 3}}) {
            Polynomial poly(elements_of(c, field), field);
            std::vector<FieldElement> values;
            for (auto &x : domain) values.push_back(poly.eval(x));
            Polynomial got = Polynomial::interpolate_domain(domain, values);
            EXPECT(got == poly);
            EXPECT(got.coeffs.size() == 4);
        }
        EXPECT(Polynomial::interpolate_domain(domain, elements_of({0, 0, 0, 0}, field)).coeffs.empty());
        Polynomial q(elements_of({1, 2, 3}, field), field);
        Polynomial sc = q.scale(field.new_element(2));   // mod.rs:439-456
        EXPECT(sc.coeffs[0].value == 1 && sc.coeffs[1].value == 4 && sc.coeffs[2].value == 12);
        EXPECT(panics_with([&] { Polynomial::interpolate_domain(elements_of({1, 2, 3, 4}, field), elements_of({1, 4, 9, 16}, field)); },
                           "not offset"));
    }

    // src/merkle.rs:104-133
    {
        std::vector<Hash> leaves;
        for (uint8_t i = 0; i < 8; i++) leaves.push_back(Hash::from_bytes(&i, 1));
        MerkleTree tree(leaves);
        for (size_t i = 0; i < leaves.size(); i++) EXPECT(MerkleTree::verify(leaves[i], i, tree.open(i), tree.get_root()));
        uint8_t wrong = 99;
        EXPECT(!MerkleTree::verify(Hash::from_bytes(&wrong, 1), 0, tree.open(0), tree.get_root()));
        EXPECT(MerkleTree::commit(leaves) == tree.get_root());
        EXPECT(panics_with([&] { std::vector<Hash> three(leaves.begin(), leaves.begin() + 3); MerkleTree t3(three); },
                           "Number of leaves must be power of 2"));
        EXPECT(panics_with([&] { tree.open(8); }, "Index out of bounds"));
        EXPECT(Hash::from_u64(5) == Hash::from_field_elements({5}));
    }

    // src/fri.rs:533-693
    fri_case(32, 4, 2, 3, {5});
    fri_case(64, 4, 3, 7, {5, 3});
    fri_case(128, 4, 4, 13, {1, 3, 2});
    fri_case(256, 8, 5, 17, {1, 2, 5, 3, 7, 4, 1, 2});
    EXPECT(panics_with([&] { Fri(field.prim_nth_root(64), field.new_element(3), 48, 4, 2); }, "Domain length must be power of 2"));
    EXPECT(panics_with([&] { Fri(field.prim_nth_root(64), field.new_element(3), 64, 2, 2); }, "Expansion factor must be at least 4"));

    // trace.rs fibonacci + build-defined LDE: offset 1 reproduces the trace at stride 8
    {
        Trace tr = Trace::fibonacci(64);
        auto ext = tr.lde(field, 3, 1);
        bool same = true;
        __int128 a = 1, b = 1;
        for (size_t i = 0; i < 64; i++) {
            same = same && ext[0][8 * i] == (uint64_t)(a % P);
            __int128 nx = a + b; a = b; b = nx;
        }
        EXPECT(same);
    }
    // multi-GPU wrapper over a one-rank RCCL communicator: the sharded Fri::prove must give the same
    // ProofStream objects as Fri::prove on the host-buffer path (src/fri.rs:250-311)
    {
        const size_t n = 256;
        FieldElement omega = field.prim_nth_root(n), offset = field.new_element(3);
        Fri fri(omega, offset, n, 8, 5);
        std::vector<uint64_t> cw(n);
        for (size_t i = 0; i < n; i++) cw[i] = (i * i + 7) % P;
        ProofStream ps_host;
        FiatShamir fs;
        std::vector<size_t> top_host = fri.prove(elements_of(cw, field), fs, ps_host);
        uint32_t *d_cw = nullptr;
        check(smi_dev_alloc(field.ctx(), n * 4, (void **)&d_cw), field.ctx());
        check(smi_dev_upload_u64(field.ctx(), cw.data(), n, d_cw, 0), field.ctx());
        MultiGpu mg(field, MultiGpu::unique_id(), 0, 1);
        ProofStream ps_dev;
        smi_fri_cfg cfg{omega.value, offset.value, n, 8, 5};
        std::vector<size_t> top_dev = mg.fri_prove(cfg, d_cw, n, ps_dev);
        EXPECT(top_dev == top_host);
        EXPECT(ps_dev.serialize() == ps_host.serialize());
        check(smi_dev_free(field.ctx(), d_cw), field.ctx());
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
