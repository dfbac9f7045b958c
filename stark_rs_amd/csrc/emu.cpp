// emu.cpp -- CPU emulator of the NTT kernels' thread phases (TEST INFRASTRUCTURE).
//
// The build container has no GPU, so the non-GPU tests run the very same phase functions
// (ntt_core.h) and pass sequencing (ntt_driver.h) the HIP kernels use, one "thread" at a
// time, and compare with the oracle.  This catches indexing / twiddle / planning mistakes
// before GPU time is spent.  The product never loads this library: stark_rs_amd fails
// loudly without libstarkmi.so and a GPU.
#include <array>
#include <vector>

#include "ntt_driver.h"
#include "lde_core.h"
#include "tables.h"
#include "proof_parse.h"

namespace {

std::vector<uint32_t> fill(const GeomSpec &s, const Fp &F) {
    std::vector<uint32_t> t((size_t)s.count << s.pair);
    for (uint32_t i = 0; i < s.count; i++) {
        const uint32_t v = geom_entry(s.c_m, s.q_m, i, s.stride, F);
        if (s.pair) { const Tw2 c = tw2_from_mont(v, F); t[2 * i] = c.w; t[2 * i + 1] = c.q; }
        else t[i] = v;
    }
    return t;
}

// ntt_pass_cols_kernel: every column of a tile by one workgroup, output multipliers derived once per thread
template <int LOGR, int LOGW, int KIND, int CAP> void emu_pass_cols(const PassArgs &a) {
    typedef NttPass<LOGR, LOGW, KIND, CAP> NP;
    constexpr bool LAST = KIND == PASS_LAST;
    const uint32_t NT = NP::NT;
    std::vector<uint32_t> tile(NP::R * NP::WP);
    std::vector<Tw2> tw(NP::R);
    std::vector<std::array<uint32_t, 16>> mw(NT), mq(NT), iw(NT), regs(NT);
    const bool twin = KIND == PASS_MID && (a.flags & NTT_TW_IN);   // ntt_pass_cols_kernel<..., TWIN>
    auto r16 = [](std::array<uint32_t, 16> &x) -> uint32_t(&)[16] { return reinterpret_cast<uint32_t(&)[16]>(x); };
    for (uint32_t blk = 0; blk < a.n_tiles; blk++) {
        typename NP::TileId t = NP::tile_id(a, blk);
        for (uint32_t tid = 0; tid < NT; tid++) NP::load_tw(a, tw.data(), tid);
        for (uint32_t b = 0; b < a.batch; b++) {     // the column groups (grid.y) run the same program one after the other
            if (b % NP::COLS_PER_WG == 0)            // a new workgroup: its threads derive their multipliers again
                for (uint32_t tid = 0; tid < NT; tid++) {
                    NP::out_mul(a, t, tid, r16(mw[tid]), r16(mq[tid]));
                    if (twin) NP::in_mul(a, t, tid, r16(iw[tid]));
                }
            if (twin) {
                for (uint32_t tid = 0; tid < NT; tid++) {
                    NP::template load_regs<0, false>(a, t, b, r16(regs[tid]), tid);
                    for (int i = 0; i < 16; i++) regs[tid][i] = mont_mul(regs[tid][i], iw[tid][i], a.F);
                }
                for (uint32_t tid = 0; tid < NT; tid++) NP::template step0_regs<0>(a, r16(regs[tid]), tile.data(), tw.data(), tid);
            } else if (!LAST) {
#define ZCASE(Z)                                                                                                 \
    case Z:                                                                                                      \
        for (uint32_t tid = 0; tid < NT; tid++) NP::template load_regs<Z>(a, t, b, r16(regs[tid]), tid);         \
        for (uint32_t tid = 0; tid < NT; tid++) NP::template step0_regs<Z>(a, r16(regs[tid]), tile.data(), tw.data(), tid); \
        break;
                switch (a.zlog) { ZCASE(2) ZCASE(3) ZCASE(4) default: ZCASE(0) }
#undef ZCASE
            } else if (a.flags & NTT_LAST_DIRECT) {
                for (uint32_t tid = 0; tid < NT; tid++) NP::load_rows_direct(a, t, b, r16(regs[tid]), tid);
                for (uint32_t tid = 0; tid < NT; tid++) NP::step0_rows(a, r16(regs[tid]), tile.data(), tw.data(), tid);
            } else {
                for (uint32_t tid = 0; tid < NT; tid++) NP::load_rows(a, t, b, r16(regs[tid]), tid);
                for (uint32_t tid = 0; tid < NT; tid++) NP::rows_to_lds(r16(regs[tid]), tile.data(), tid);
                for (uint32_t tid = 0; tid < NT; tid++) NP::step0_lds(a, tile.data(), tw.data(), tid);
            }
            for (uint32_t tid = 0; tid < NT; tid++) NP::step_mid(a, tile.data(), tw.data(), tid);
            for (uint32_t tid = 0; tid < NT; tid++) NP::last_step_store_mul(a, t, b, tile.data(), tw.data(), tid, r16(mw[tid]), r16(mq[tid]));
        }
    }
}

// emu_set_share_cols: 1 = the launcher's rule (HipLauncher::launch), 0 = never, 2 = whenever the pass has multipliers
// (the tests' sizes are far below the launch-size threshold of the rule)
int g_share_cols = 1;

template <int LOGR, int LOGW, int KIND, int CAP> void emu_pass(const PassArgs &a, uint32_t batch) {
    typedef NttPass<LOGR, LOGW, KIND, CAP> NP;
    constexpr bool LAST = KIND == PASS_LAST;
    if (g_share_cols && NP::share_cols(a, g_share_cols == 2 ? 0u : (uint32_t)NP::COLS_MIN_WGS)) return emu_pass_cols<LOGR, LOGW, KIND, CAP>(a);
    std::vector<uint32_t> tile(NP::R * NP::WP);
    std::vector<Tw2> tw(NP::R);
    for (uint32_t b = 0; b < batch; b++)
        for (uint32_t blk = 0; blk < a.n_tiles; blk++) {
            typename NP::TileId t = NP::tile_id(a, blk);
            const uint32_t NT = NP::NT;
            for (uint32_t tid = 0; tid < NT; tid++) NP::load_tw(a, tw.data(), tid);
            if (!LAST) {
                std::vector<std::array<uint32_t, 16>> regs(NT);   // per-thread registers live across the barrier
#define ZCASE(Z)                                                                                                           \
    case Z:                                                                                                                \
        for (uint32_t tid = 0; tid < NT; tid++) NP::template load_regs<Z>(a, t, b, reinterpret_cast<uint32_t(&)[16]>(regs[tid]), tid); \
        for (uint32_t tid = 0; tid < NT; tid++) NP::template step0_regs<Z>(a, reinterpret_cast<uint32_t(&)[16]>(regs[tid]), tile.data(), tw.data(), tid); \
        break;
                switch (a.zlog) { ZCASE(0) ZCASE(1) ZCASE(2) ZCASE(3) ZCASE(4) }
#undef ZCASE
            } else if (a.flags & NTT_LAST_DIRECT) {
                std::vector<std::array<uint32_t, 16>> regs(NT);
                for (uint32_t tid = 0; tid < NT; tid++) NP::load_rows_direct(a, t, b, reinterpret_cast<uint32_t(&)[16]>(regs[tid]), tid);
                for (uint32_t tid = 0; tid < NT; tid++) NP::step0_rows(a, reinterpret_cast<uint32_t(&)[16]>(regs[tid]), tile.data(), tw.data(), tid);
            } else {
                for (uint32_t tid = 0; tid < NT; tid++) {
                    uint32_t v[16];
                    NP::load_rows(a, t, b, v, tid);
                    NP::rows_to_lds(v, tile.data(), tid);
                }
                for (uint32_t tid = 0; tid < NT; tid++) NP::step0_lds(a, tile.data(), tw.data(), tid);
            }
            for (uint32_t tid = 0; tid < NT; tid++) NP::step_mid(a, tile.data(), tw.data(), tid);
            for (uint32_t tid = 0; tid < NT; tid++) NP::last_step_store(a, t, b, tile.data(), tw.data(), tid);
        }
}

struct EmuLauncher {
    void small(const SmallArgs &a, uint32_t batch) {
        std::vector<uint32_t> buf((1u << a.L) + ((1u << a.L) >> 6) + 1);
        std::vector<Tw2> twm(((1u << a.L) >> 1) + 1);
        for (uint32_t tid = 0; tid < SMI_NTT_THREADS; tid++) NttSmall::load_tw(a, twm.data(), tid);
        for (uint32_t b = 0; b < batch; b++) {
            for (uint32_t tid = 0; tid < SMI_NTT_THREADS; tid++) NttSmall::load(a, b, buf.data(), tid);
            uint32_t s = 0;
            for (; s + 2 <= a.L; s += 2)
                for (uint32_t tid = 0; tid < SMI_NTT_THREADS; tid++) NttSmall::stage4(a, s, buf.data(), twm.data(), tid);
            if (s < a.L)
                for (uint32_t tid = 0; tid < SMI_NTT_THREADS; tid++) NttSmall::stage(a, s, buf.data(), twm.data(), tid);
            for (uint32_t tid = 0; tid < SMI_NTT_THREADS; tid++) NttSmall::store(a, b, buf.data(), tid);
        }
    }
    void pass(int logr, int logw, bool last, const PassArgs &a, uint32_t batch) {
#define X(LR, LW)                                           \
    if (logr == LR && logw == LW) {                         \
        const bool wide = a.F.p < (1u << 29);               \
        const int kind = last ? PASS_LAST : (a.flags & NTT_FIRST) ? PASS_FIRST : PASS_MID;   \
        if (wide && kind == PASS_FIRST) emu_pass<LR, LW, PASS_FIRST, 8>(a, batch);   \
        else if (wide && kind == PASS_MID) emu_pass<LR, LW, PASS_MID, 8>(a, batch);  \
        else if (wide) emu_pass<LR, LW, PASS_LAST, 8>(a, batch);                     \
        else if (kind == PASS_FIRST) emu_pass<LR, LW, PASS_FIRST, 4>(a, batch);      \
        else if (kind == PASS_MID) emu_pass<LR, LW, PASS_MID, 4>(a, batch);          \
        else emu_pass<LR, LW, PASS_LAST, 4>(a, batch);                               \
        return;                                             \
    }
        SMI_NTT_SHAPES(X)
#undef X
    }
};

}  // namespace

static bool g_emu_last_direct = true;    // the product's default (internal.h)
extern "C" void emu_set_last_direct(int on) { g_emu_last_direct = on != 0; }
static int g_emu_defer_tw = 2;   // NttRequest::defer_tw: 0 never, 1 always, 2 the driver's rule
extern "C" void emu_set_defer_tw(int mode) { g_emu_defer_tw = mode; }
extern "C" void emu_set_share_cols(int mode) { g_share_cols = mode; }
extern "C" int emu_ntt(uint64_t p, uint64_t g, const uint32_t *in, uint32_t *out, uint32_t L, uint32_t n_in,
                       uint32_t batch, uint64_t in_stride, uint64_t out_stride, int inverse, uint64_t offset,
                       uint64_t post_scale) {
    FieldSetup fs;
    if (!field_setup(p, g, &fs) || L > fs.K) return -1;
    const Fp &F = fs.F;
    GeomSpec sp[3];
    ntt_table_specs(fs, inverse, sp);
    std::vector<uint32_t> tw10 = fill(sp[0], F), lo = fill(sp[1], F), hi = fill(sp[2], F);
    NttRequest rq;
    memset(&rq, 0, sizeof rq);
    rq.T = NttTables{(const Tw2 *)tw10.data(), lo.data(), hi.data(), fs.K, ntt_table_h(fs.K)};
    std::vector<uint32_t> slo, shi;
    const uint32_t pp = F.p;
    if (!inverse) {
        rq.pre_scale = offset % pp != 1;
        if (rq.pre_scale) {
            GeomSpec s2[2];
            rq.q_plain = (uint32_t)(offset % pp);
            scale_table_specs(F, 1, (uint32_t)(offset % pp), L, s2);
            slo = fill(s2[0], F); shi = fill(s2[1], F);
        }
    } else {
        rq.post_scale = true;
        const uint32_t ninv = host_powmod((uint32_t)((1ull << L) % pp), pp - 2, pp);
        const uint32_t q = host_mulmod((uint32_t)(post_scale % pp), host_powmod((uint32_t)(offset % pp), pp - 2, pp), pp);
        GeomSpec s2[2];
        rq.q_plain = q;
        scale_table_specs(F, ninv, q, L, s2);
        slo = fill(s2[0], F); shi = fill(s2[1], F);
    }
    rq.S = ScaleTables{slo.data(), shi.data(), scale_table_h(L)};
    std::vector<uint32_t> scratch((size_t)batch << L);
    rq.in = in; rq.out = out; rq.scratch = scratch.data();
    rq.L = L; rq.n_in = n_in; rq.batch = batch; rq.in_stride = in_stride; rq.out_stride = out_stride; rq.F = F;
    rq.defer_tw = g_emu_defer_tw;
    rq.last_direct = g_emu_last_direct;
    EmuLauncher ln;
    return ntt_run(ln, rq) ? 0 : -1;
}

// ---- one transform sharded over ranks (ntt_driver.h): this rank's two local phases
namespace {
struct EmuTables {
    std::vector<uint32_t> tw10, lo, hi, slo, shi;
    NttRequest rq;
};
bool emu_shard_request(uint64_t p, uint64_t g, uint32_t L, int inverse, uint64_t offset, EmuTables &t) {
    FieldSetup fs;
    if (!field_setup(p, g, &fs) || L > fs.K) return false;
    const Fp &F = fs.F;
    GeomSpec sp[3], s2[2];
    ntt_table_specs(fs, inverse, sp);
    t.tw10 = fill(sp[0], F); t.lo = fill(sp[1], F); t.hi = fill(sp[2], F);
    memset(&t.rq, 0, sizeof t.rq);
    t.rq.T = NttTables{(const Tw2 *)t.tw10.data(), t.lo.data(), t.hi.data(), fs.K, ntt_table_h(fs.K)};
    t.rq.L = L; t.rq.n_in = 1u << L; t.rq.batch = 1; t.rq.F = F;
    if (!inverse) {
        t.rq.pre_scale = offset % F.p != 1;
        t.rq.q_plain = (uint32_t)(offset % F.p);
        scale_table_specs(F, 1, t.rq.q_plain, L, s2);
    } else {
        t.rq.post_scale = true;
        t.rq.q_plain = 1;
        scale_table_specs(F, host_powmod((uint32_t)((1ull << L) % F.p), F.p - 2, F.p), 1, L, s2);
    }
    t.slo = fill(s2[0], F); t.shi = fill(s2[1], F);
    t.rq.S = ScaleTables{t.slo.data(), t.shi.data(), scale_table_h(L)};
    return true;
}
}  // namespace
extern "C" int emu_ntt_shard_first(uint64_t p, uint64_t g, uint32_t *strip, uint32_t L, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset) {
    EmuTables t;
    if (!emu_shard_request(p, g, L, inverse, offset, t)) return -1;
    t.rq.in = strip; t.rq.out = strip;
    EmuLauncher ln;
    return ntt_run_shard_first(ln, t.rq, log_g, rank) ? 0 : -1;
}
extern "C" int emu_ntt_shard_rest(uint64_t p, uint64_t g, uint32_t *rows, uint32_t *out, uint32_t L, uint32_t log_g, int inverse) {
    EmuTables t;
    if (!emu_shard_request(p, g, L, inverse, 1, t)) return -1;
    t.rq.scratch = rows; t.rq.out = out;
    t.rq.last_direct = g_emu_last_direct;
    EmuLauncher ln;
    return ntt_run_shard_rest(ln, t.rq, log_g) ? 0 : -1;
}

// ---- two-pass low-degree extension (lde_core.h): the kernels' phases, one "thread" at a time
template <int LOGR, int CAP> void emu_lde_a(const LdeArgs &a) {
    typedef LdeA<LOGR, CAP> A;
    std::vector<uint32_t> tile(A::R * 4);
    std::vector<Tw2> tw(A::TWS);
    for (uint32_t b = 0; b < a.batch; b++)
        for (uint32_t blk = 0; blk < a.n_tiles; blk++) {
            const typename A::TileId t = A::tile_id(a, blk);
            std::vector<std::array<uint32_t, 16>> regs(A::NT);
            for (uint32_t tid = 0; tid < A::NT; tid++) A::load_tw(a, tw.data(), tid);
            for (uint32_t tid = 0; tid < A::NT; tid++) A::load_regs(a, t, b, reinterpret_cast<uint32_t(&)[16]>(regs[tid]), tid);
            for (uint32_t tid = 0; tid < A::NT; tid++) A::step0(a, reinterpret_cast<uint32_t(&)[16]>(regs[tid]), tile.data(), tw.data(), tid);
            for (uint32_t tid = 0; tid < A::NT; tid++) A::step_mid(a, tile.data(), tw.data(), tid);
            for (uint32_t tid = 0; tid < A::NT; tid++) A::last_step_store(a, t, b, tile.data(), tw.data(), tid);
        }
}
template <int CAP> void emu_lde_b(const LdeArgs &a) {
    typedef LdeB<CAP> B;
    std::vector<uint32_t> tile(B::R * B::WP);
    std::vector<Tw2> tw(B::R);
    for (uint32_t b = 0; b < a.batch; b++)
        for (uint32_t blk = 0; blk < a.n_tiles; blk++) {
            const typename B::TileId t = B::tile_id(a, blk);
            for (uint32_t tid = 0; tid < B::NT; tid++) B::load_tw(a, tw.data(), tid);
            for (uint32_t tid = 0; tid < B::NT; tid++) {
                uint32_t v[16];
                B::load(a, t, b, v, tid);
                B::to_lds(a, v, tile.data(), tid);
            }
            for (uint32_t tid = 0; tid < B::NT; tid++) B::step0(a, t, tile.data(), tw.data(), tid);
            for (uint32_t tid = 0; tid < B::NT; tid++) B::step_mid(a, tile.data(), tw.data(), tid);
            for (uint32_t tid = 0; tid < B::NT; tid++) B::last_step_store(a, t, b, tile.data(), tw.data(), tid);
        }
}
// coef: batch columns of 2^L coefficients (stride 2^L); out: batch columns of 2^(L+beta) evaluations
// on the subgroup of that order (offset 1: the coset shift is applied to the coefficients upstream)
static int g_emu_lde_geo = -1, g_emu_lde_lay = -1;
extern "C" void emu_lde2_knobs(int geo_rq, int lay_kq) { g_emu_lde_geo = geo_rq; g_emu_lde_lay = lay_kq; }
extern "C" int emu_lde2(uint64_t p, uint64_t g, const uint32_t *coef, uint32_t *out, uint32_t L, uint32_t beta, uint32_t batch) {
    FieldSetup fs;
    if (!field_setup(p, g, &fs) || L + beta > fs.K || !lde2_supported(L, beta)) return -1;
    const Fp &F = fs.F;
    GeomSpec sp[3];
    ntt_table_specs(fs, 0, sp);
    std::vector<uint32_t> tw10 = fill(sp[0], F), lo = fill(sp[1], F), hi = fill(sp[2], F);
    std::vector<uint32_t> mid((size_t)batch << (L + beta)), coef_t((size_t)batch << L);
    const uint32_t log_m = L - SMI_LDE_LOGB + beta;
    const uint32_t wm = host_powmod(fs.wmax[0], 1ull << (fs.K - log_m), F.p);
    std::vector<uint32_t> ctab = fill(GeomSpec{F.r1, (uint32_t)(((uint64_t)wm << 32) % F.p), 1, 1u << log_m, 1}, F);
    LdeArgs a;
    memset(&a, 0, sizeof a);
    a.coef = coef; a.mid = mid.data(); a.coef_t = coef_t.data(); a.ctab = (const Tw2 *)ctab.data(); a.out = out; a.coef_stride = 1ull << L; a.out_stride = 1ull << (L + beta);
    a.F = F; a.T = NttTables{(const Tw2 *)tw10.data(), lo.data(), hi.data(), fs.K, ntt_table_h(fs.K)};
    a.L = L; a.beta = beta; a.batch = batch;
    a.geo_rq = g_emu_lde_geo >= 0 && (uint32_t)g_emu_lde_geo <= beta ? (uint32_t)g_emu_lde_geo : lde_default_geo_rq(beta);
    a.lay_kq = g_emu_lde_lay >= (int)(SMI_LDE_BLINES_LOG - a.geo_rq) && g_emu_lde_lay <= 4 ? (uint32_t)g_emu_lde_lay : SMI_LDE_BLINES_LOG - a.geo_rq;
    const bool wide = F.p < (1u << 29);
    const int logr = (int)L - SMI_LDE_LOGB;
    {
        std::vector<uint32_t> tile(LdeCoefTile::T * (LdeCoefTile::T + 1) * 4);
        for (uint32_t b = 0; b < batch; b++)
            for (uint32_t by = 0; by < (1u << logr) / LdeCoefTile::T; by++)
                for (uint32_t bx = 0; bx < 256 / LdeCoefTile::T; bx++) {
                    for (uint32_t tid = 0; tid < LdeCoefTile::NT; tid++) LdeCoefTile::load(a, bx, by, b, tile.data(), tid);
                    for (uint32_t tid = 0; tid < LdeCoefTile::NT; tid++) LdeCoefTile::store(a, bx, by, b, tile.data(), tid);
                }
    }
    a.n_tiles = 256u << beta;
#define LA(LR) if (logr == LR) { if (wide) emu_lde_a<LR, 8>(a); else emu_lde_a<LR, 4>(a); }
    LA(10) LA(11) LA(12)
#undef LA
    a.n_tiles = 1u << (L + beta - SMI_LDE_LOGB - SMI_LDE_BLINES_LOG);
    if (wide) emu_lde_b<8>(a); else emu_lde_b<4>(a);
    return 0;
}

// ---- hash phases (hash_core.h) --------------------------------------------------------
#include "hash_core.h"
// pairs go through the two-hashes-per-state path the Merkle kernels use, the odd tail through the
// single-hash path
extern "C" void emu_leaf_hash(const uint32_t *v, size_t n, uint8_t *out) {
    size_t i = 0;
    for (; i + 1 < n; i += 2) {
        uint32_t d0[8], d1[8];
        hashc::leaf_hash2(v[i], v[i + 1], d0, d1);
        memcpy(out + 32 * i, d0, 32);
        memcpy(out + 32 * (i + 1), d1, 32);
    }
    for (; i < n; i++) {
        uint32_t d[8];
        hashc::leaf_hash(v[i], d);
        memcpy(out + 32 * i, d, 32);
    }
}
extern "C" void emu_node_hash(const uint8_t *pairs, size_t n, uint8_t *out) {
    size_t i = 0;
    for (; i + 1 < n; i += 2) {
        uint32_t l0[8], r0[8], l1[8], r1[8], d0[8], d1[8];
        memcpy(l0, pairs + 64 * i, 32);
        memcpy(r0, pairs + 64 * i + 32, 32);
        memcpy(l1, pairs + 64 * (i + 1), 32);
        memcpy(r1, pairs + 64 * (i + 1) + 32, 32);
        hashc::node_hash2(l0, r0, l1, r1, d0, d1);
        memcpy(out + 32 * i, d0, 32);
        memcpy(out + 32 * (i + 1), d1, 32);
    }
    for (; i < n; i++) {
        uint32_t l[8], r[8], d[8];
        memcpy(l, pairs + 64 * i, 32);
        memcpy(r, pairs + 64 * i + 32, 32);
        hashc::node_hash(l, r, d);
        memcpy(out + 32 * i, d, 32);
    }
}
// ---- one hash over a quad of lanes (hash_quad.h): the device code itself, with `word` = four lanes
// stepped in lockstep and the DPP quad_perm move as a permutation of those four values
struct Q4 {
    uint32_t v[4];
    Q4() : v{0, 0, 0, 0} {}
    Q4(uint32_t x) : v{x, x, x, x} {}
    Q4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) : v{a, b, c, d} {}
};
#define Q4_BIN(op)                                                 \
    inline Q4 operator op(const Q4 &a, const Q4 &b) {              \
        Q4 r;                                                      \
        for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i];     \
        return r;                                                  \
    }
Q4_BIN(+) Q4_BIN(^) Q4_BIN(&) Q4_BIN(|) Q4_BIN(*)
#undef Q4_BIN
inline Q4 operator<<(const Q4 &a, int sh) { return Q4(a.v[0] << sh, a.v[1] << sh, a.v[2] << sh, a.v[3] << sh); }
inline Q4 operator>>(const Q4 &a, int sh) { return Q4(a.v[0] >> sh, a.v[1] >> sh, a.v[2] >> sh, a.v[3] >> sh); }
namespace hashc {
#define Q4_MAP3(name)                                                          \
    inline Q4 name(const Q4 &a, const Q4 &b, const Q4 &c) {                    \
        Q4 r;                                                                  \
        for (int i = 0; i < 4; i++) r.v[i] = name(a.v[i], b.v[i], c.v[i]);     \
        return r;                                                              \
    }
Q4_MAP3(pk_mad_u16) Q4_MAP3(bfi32) Q4_MAP3(xor3) Q4_MAP3(add3)
#undef Q4_MAP3
inline Q4 dup_hi16(const Q4 &a) { return Q4(dup_hi16(a.v[0]), dup_hi16(a.v[1]), dup_hi16(a.v[2]), dup_hi16(a.v[3])); }
inline Q4 funnel16(const Q4 &hi, const Q4 &lo) {
    return Q4(funnel16(hi.v[0], lo.v[0]), funnel16(hi.v[1], lo.v[1]), funnel16(hi.v[2], lo.v[2]), funnel16(hi.v[3], lo.v[3]));
}
inline Q4 perm8(const Q4 &hi, const Q4 &lo, uint32_t sel) {
    return Q4(perm8(hi.v[0], lo.v[0], sel), perm8(hi.v[1], lo.v[1], sel), perm8(hi.v[2], lo.v[2], sel), perm8(hi.v[3], lo.v[3], sel));
}
inline void absorb32_words(Q4 P[8], const Q4 M[8]) {   // every lane runs the whole absorb, as on the device
    for (int i = 0; i < 4; i++) {
        uint32_t p[8], m[8];
        for (int j = 0; j < 8; j++) { p[j] = P[j].v[i]; m[j] = M[j].v[i]; }
        absorb32_words(p, m);
        for (int j = 0; j < 8; j++) P[j].v[i] = p[j];
    }
}
}  // namespace hashc
#define SMI_QUAD_EMU
namespace hashq {
#define SMI_QD inline
typedef Q4 word;
template <int P0, int P1, int P2, int P3> inline Q4 quad(const Q4 &x) { return Q4(x.v[P0], x.v[P1], x.v[P2], x.v[P3]); }
inline Q4 lane_in_quad(uint32_t) { return Q4(0, 1, 2, 3); }
inline Q4 mask_ge(const Q4 &q, uint32_t k) { return Q4(q.v[0] >= k ? ~0u : 0u, q.v[1] >= k ? ~0u : 0u, q.v[2] >= k ? ~0u : 0u, q.v[3] >= k ? ~0u : 0u); }
inline Q4 mask_eq(const Q4 &q, uint32_t k) { return Q4(q.v[0] == k ? ~0u : 0u, q.v[1] == k ? ~0u : 0u, q.v[2] == k ? ~0u : 0u, q.v[3] == k ? ~0u : 0u); }
inline Q4 sel4(const Q4 &q, const Q4 &a, const Q4 &b, const Q4 &c, const Q4 &d) {
    Q4 r;
    for (int i = 0; i < 4; i++) r.v[i] = q.v[i] == 0 ? a.v[i] : q.v[i] == 1 ? b.v[i] : q.v[i] == 2 ? c.v[i] : d.v[i];
    return r;
}
}  // namespace hashq
#include "hash_quad.h"
extern "C" void emu_node_hash_quad(const uint8_t *pairs, size_t n, uint8_t *out) {
    const hashq::Lane L = hashq::make_lane(0);
    for (size_t i = 0; i < n; i++) {
        uint32_t l[8], r[8], d[8];
        memcpy(l, pairs + 64 * i, 32);
        memcpy(r, pairs + 64 * i + 32, 32);
        Q4 ql[8], qr[8], lo, hi;
        for (int j = 0; j < 8; j++) { ql[j] = Q4(l[j]); qr[j] = Q4(r[j]); }
        hashq::node_hash(ql, qr, L, lo, hi);
        for (int q = 0; q < 4; q++) { d[q] = lo.v[q]; d[4 + q] = hi.v[q]; }
        memcpy(out + 32 * i, d, 32);
    }
}

// ---- one hash over a row of sixteen lanes (hash_hex.h): the device code itself, with `word` = sixteen lanes stepped in
// lockstep and the DPP row moves as permutations of those sixteen values
struct H16 {
    uint32_t v[16];
    H16() { for (int i = 0; i < 16; i++) v[i] = 0; }
    H16(uint32_t x) { for (int i = 0; i < 16; i++) v[i] = x; }
};
#define H16_BIN(op)                                                \
    inline H16 operator op(const H16 &a, const H16 &b) {           \
        H16 r;                                                     \
        for (int i = 0; i < 16; i++) r.v[i] = a.v[i] op b.v[i];    \
        return r;                                                  \
    }
H16_BIN(+) H16_BIN(^) H16_BIN(&) H16_BIN(|) H16_BIN(*)
#undef H16_BIN
inline H16 operator<<(const H16 &a, int sh) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = a.v[i] << sh; return r; }
inline H16 operator>>(const H16 &a, int sh) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = a.v[i] >> sh; return r; }
namespace hashc {
#define H16_MAP3(name)                                                         \
    inline H16 name(const H16 &a, const H16 &b, const H16 &c) {                \
        H16 r;                                                                 \
        for (int i = 0; i < 16; i++) r.v[i] = name(a.v[i], b.v[i], c.v[i]);    \
        return r;                                                              \
    }
H16_MAP3(pk_mad_u16) H16_MAP3(bfi32) H16_MAP3(xor3)
#undef H16_MAP3
inline H16 funnel16(const H16 &hi, const H16 &lo) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = funnel16(hi.v[i], lo.v[i]); return r; }
}  // namespace hashc
#define SMI_HEX_EMU
namespace hashx {
#define SMI_XD inline
typedef H16 word;
template <int N> inline H16 rot(const H16 &x) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = x.v[(i - N) & 15]; return r; }
template <int N> inline H16 shr(const H16 &x) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = i >= N ? x.v[i - N] : 0u; return r; }
template <int P0, int P1, int P2, int P3> inline H16 quad(const H16 &x) {
    const int P[4] = {P0, P1, P2, P3};
    H16 r;
    for (int i = 0; i < 16; i++) r.v[i] = x.v[(i & ~3) | P[i & 3]];
    return r;
}
inline H16 lane_in_row(uint32_t) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = (uint32_t)i; return r; }
inline H16 mask_range(const H16 &w, uint32_t lo, uint32_t hi) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = w.v[i] >= lo && w.v[i] < hi ? ~0u : 0u; return r; }
inline H16 table16(const H16 &w, const uint32_t (&t)[16]) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = t[w.v[i] & 15]; return r; }
inline H16 perm8v(const H16 &hi, const H16 &lo, const H16 &sel) { H16 r; for (int i = 0; i < 16; i++) r.v[i] = hashc::perm8(hi.v[i], lo.v[i], sel.v[i]); return r; }
inline H16 lshl_or(const H16 &a, int sh, const H16 &b) { return (a << sh) | b; }
inline H16 lshl_add(const H16 &a, int sh, const H16 &b) { return (a << sh) + b; }
}  // namespace hashx
#include "hash_hex.h"
extern "C" void emu_node_hash_hex(const uint8_t *pairs, size_t n, uint8_t *out) {
    const hashx::Lane L = hashx::make_lane(0);
    for (size_t i = 0; i < n; i++) {
        uint32_t l[8], r[8];
        memcpy(l, pairs + 64 * i, 32);
        memcpy(r, pairs + 64 * i + 32, 32);
        H16 llo, lhi, rlo, rhi;   // what each lane loads: natural words w >> 2 and 4 + (w >> 2) of either child
        for (int w = 0; w < 16; w++) { llo.v[w] = l[w >> 2]; lhi.v[w] = l[4 + (w >> 2)]; rlo.v[w] = r[w >> 2]; rhi.v[w] = r[4 + (w >> 2)]; }
        const H16 x = hashx::node_hash(hashx::message(llo, lhi, L), hashx::message(rlo, rhi, L), L);
        for (int w = 0; w < 16; w++) { out[32 * i + w] = (uint8_t)x.v[w]; out[32 * i + 16 + w] = (uint8_t)(x.v[w] >> 16); }
    }
}

// one Fiat-Shamir round over the row (hashx::fs_absorb / fs_challenge): state in, state and challenge out
extern "C" void emu_fs_round_hex(uint32_t *fs_words, const uint8_t *root, uint64_t *alpha) {
    const hashx::Lane L = hashx::make_lane(0);
    uint32_t r[8];
    memcpy(r, root, 32);
    H16 lo, hi, x;
    for (int w = 0; w < 16; w++) { lo.v[w] = r[w >> 2]; hi.v[w] = r[4 + (w >> 2)]; x.v[w] = fs_words[w]; }
    x = hashx::fs_absorb(x, hashx::message(lo, hi, L), L);
    for (int w = 0; w < 16; w++) fs_words[w] = x.v[w];
    const H16 y = hashx::fs_challenge(x, L);
    *alpha = 0;
    for (int k = 0; k < 8; k++) *alpha |= (uint64_t)(y.v[k] & 0xFFu) << (8 * k);
}
// the same round by a single lane (hashc::fs_absorb_root, what the row form replaces)
extern "C" void emu_fs_round(uint32_t *fs_words, const uint8_t *root, uint64_t *alpha) {
    uint32_t m[8];
    memcpy(m, root, 32);
    hashc::fs_absorb_root(fs_words, m, nullptr, alpha);
}

// rows of W residues, row-major in `v`: pairs through row_hash2 when W <= 4, the rest through row_hash
extern "C" void emu_row_hash(const uint32_t *v, size_t n_rows, int W, uint8_t *out) {
    size_t i = 0;
    if (W >= 1 && W <= 4)
        for (; i + 1 < n_rows; i += 2) {
            uint32_t d0[8], d1[8];
            hashc::row_hash2(v + i * W, v + (i + 1) * W, W, d0, d1);
            memcpy(out + 32 * i, d0, 32);
            memcpy(out + 32 * (i + 1), d1, 32);
        }
    for (; i < n_rows; i++) {
        uint32_t d[8];
        hashc::row_hash(v + i * W, W, d);
        memcpy(out + 32 * i, d, 32);
    }
}
extern "C" void emu_hash_bytes(const uint8_t *msg, size_t len, uint8_t *out) {
    uint32_t d[8];
    hashc::hash_bytes(msg, len, d);
    memcpy(out, d, 32);
}

// ---- FRI fold arithmetic (fri_core.h) ---------------------------------------------------
#include "fri_core.h"
extern "C" int emu_fold_shard(uint64_t p, uint64_t g, const uint32_t *lo, const uint32_t *hi, uint32_t count, uint32_t i0,
                              uint32_t full_len, uint64_t alpha, uint64_t offset, uint64_t omega, uint32_t *out) {
    FieldSetup fs;
    if (!field_setup(p, g, &fs)) return -1;
    const Fp &F = fs.F;
    const uint32_t pp = F.p;
    uint32_t L = 0;
    while ((1u << L) < full_len / 2) L++;
    GeomSpec sp[2];
    scale_table_specs(F, host_powmod((uint32_t)offset, pp - 2, pp), host_powmod((uint32_t)omega, pp - 2, pp), L, sp);
    std::vector<uint32_t> slo = fill(sp[0], F), shi = fill(sp[1], F);
    const ScaleTables S{slo.data(), shi.data(), scale_table_h(L)};
    const uint32_t inv2_m = (uint32_t)(((uint64_t)host_powmod(2, pp - 2, pp) << 32) % pp);
    const uint32_t ah_m = fold_alpha_half(alpha, inv2_m, F);
    for (uint32_t i = 0; i < count; i++) out[i] = fold_element(lo[i], hi[i], i0 + i, ah_m, inv2_m, S, F);
    return 0;
}

// ProofStream::deserialize as smi_fri_verify runs it (csrc/proof_parse.h), exposed for the CPU fuzz / sanitizer
// tests: tags, element counts and payload offsets of up to `cap` objects; returns how many objects there are.
extern "C" size_t emu_proof_parse(const uint8_t *b, size_t n, size_t max_objs, int32_t *tags, uint64_t *counts, uint64_t *offsets,
                                  size_t cap, size_t *end) {
    size_t e = 0;
    const std::vector<proofp::Obj> objs = proofp::parse(b, n, max_objs, &e);
    for (size_t i = 0; i < objs.size() && i < cap; i++) {
        tags[i] = objs[i].tag;
        counts[i] = objs[i].count;
        offsets[i] = (uint64_t)(objs[i].p - b);
    }
    if (end) *end = e;
    return objs.size();
}
