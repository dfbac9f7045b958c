// ntt_driver.h -- sequences the passes of one batched transform.  Templated on a
// Launcher so that the HIP path (ntt.hip) and the CPU emulator of the non-GPU tests
// (emu.cpp) share every decision about digits, strides, flags and scratch use.
#pragma once
#include "ntt_host.h"

struct NttRequest {
    const uint32_t *in;
    uint32_t *out;
    uint32_t *scratch;   // batch * 2^L elements, required whenever the plan has passes (ntt_make_plan(L, batch).np > 0)
    uint32_t L;
    uint32_t n_in;       // <= 2^L; inputs beyond it read as zero
    uint32_t batch;
    uint64_t in_stride, out_stride;
    Fp F;
    NttTables T;         // direction already chosen
    ScaleTables S;
    bool pre_scale;      // multiply input i by S(i)   (coset shift of a forward transform)
    bool post_scale;     // multiply output k by S(k)  (n^-1 * offset^-k of an inverse transform)
    uint32_t q_plain;    // ratio of the scale sequence S(i) = c * q^i (plain form)
    bool last_direct;    // the last pass runs its first radix-16 step on the load registers (NTT_LAST_DIRECT): one LDS round
                         // trip fewer; 2^22 x 4 extension step 0.791 -> 0.779 ms over three A/B rounds (gpurun_out/exp_last_direct.log)
    int defer_tw;        // three or more passes: the first pass's inter-pass twiddle is applied by the second as it loads --
                         // 0 never, 1 always, 2 when the second pass takes the columns of a tile per workgroup (it then
                         // derives them once per thread for all of them, NttPass::in_mul: 2^25 x 4 first pass 199 -> 148 us,
                         // second 223 -> 250 us, gpurun_out/exp_defer_cols3.log; with one column per workgroup a wash)
};

// Launcher concept:
//   void small(const SmallArgs&, uint32_t batch);
//   void pass(int logr, int logw, bool last, const PassArgs&, uint32_t batch);
// Returns false (and launches nothing) if a multi-pass plan comes without its inter-pass buffer.
template <class Launcher> inline bool ntt_run(Launcher &ln, const NttRequest &rq) {
    const NttPlan pl = ntt_make_plan(rq.L, rq.batch);
    if (pl.np == 0) {
        SmallArgs a;
        a.in = rq.in; a.out = rq.out; a.in_stride = rq.in_stride; a.out_stride = rq.out_stride;
        a.F = rq.F; a.T = rq.T; a.S = rq.S; a.L = rq.L; a.n_in = rq.n_in;
        a.flags = (rq.pre_scale ? NTT_PRE_SCALE : 0) | (rq.post_scale ? NTT_POST_SCALE : 0);
        ln.small(a, rq.batch);
        return true;
    }
    if (!rq.scratch) return false;
    const uint64_t n = 1ull << rq.L;
    uint32_t consumed = 0;
    const uint64_t wgs1 = pl.np >= 3 ? (n >> (pl.logr[1] + pl.logw[1])) * ((rq.batch + SMI_COLS_PER_WG - 1) / SMI_COLS_PER_WG) : 0;
    const bool defer = pl.np >= 3 && (rq.defer_tw == 1 || (rq.defer_tw == 2 && rq.batch > 1 && wgs1 >= SMI_COLS_MIN_WGS));
    for (int p = 0; p < pl.np; p++) {
        const bool first = p == 0, last = p == pl.np - 1;
        PassArgs a;
        memset(&a, 0, sizeof a);
        a.in = first ? rq.in : rq.scratch;
        a.in_stride = first ? rq.in_stride : n;
        a.out = last ? rq.out : rq.scratch;
        a.out_stride = last ? rq.out_stride : n;
        a.F = rq.F; a.T = rq.T; a.S = rq.S;
        a.L = rq.L; a.Sp = consumed; a.n_in = rq.n_in;
        a.flags = (first ? NTT_FIRST : 0) | (first && rq.pre_scale ? NTT_PRE_SCALE : 0) |
                  (last && rq.post_scale ? NTT_POST_SCALE : 0);
        if (last && rq.last_direct) a.flags |= NTT_LAST_DIRECT;
        if (defer && p == 0) a.flags |= NTT_TW_SKIP;
        if (defer && p == 1) {
            a.flags |= NTT_TW_IN;
            a.prev_logr = (uint32_t)pl.logr[0];
        }
        a.d0_log = (uint32_t)pl.logr[0];
        a.n_mid = (uint32_t)(pl.np - 2);
        for (int d = 0; d < pl.np - 2; d++) a.mid_log[d] = (uint32_t)pl.logr[1 + d];
        a.n_tiles = (uint32_t)(n >> (pl.logr[p] + pl.logw[p]));
        a.batch = rq.batch;
        {   // steps of the running-product scales (see NttPass::load / store)
            const uint32_t logw = (uint32_t)pl.logw[p], blog = rq.L - consumed - (uint32_t)pl.logr[p];
            const uint64_t pre_step = (uint64_t)((1u << (pl.logr[p] + pl.logw[p] - 4)) >> logw) << blog;
            const uint32_t pr = host_powmod(rq.q_plain, pre_step, rq.F.p), po = host_powmod(rq.q_plain, 1ull << (consumed + (uint32_t)pl.logr[p] - (uint32_t)ntt_last_step_log(pl.logr[p])), rq.F.p);
            a.pre_ratio_m = (uint32_t)(((uint64_t)pr << 32) % rq.F.p);
            a.post_ratio_m = (uint32_t)(((uint64_t)po << 32) % rq.F.p);
            const uint32_t pb = host_powmod(rq.q_plain, 1ull << (consumed + (uint32_t)ntt_last_step_log(pl.logr[p])), rq.F.p);
            a.post_bi_ratio_m = (uint32_t)(((uint64_t)pb << 32) % rq.F.p);
            // zero padding seen by the first pass: rows j >= R * n_in / n hold no input
            uint32_t z = 0;
            while (first && z < 4 && ((uint64_t)rq.n_in << (z + 1)) <= n) z++;
            a.zlog = z;
        }
        ln.pass(pl.logr[p], pl.logw[p], last, a, rq.batch);
        consumed += (uint32_t)pl.logr[p];
    }
    return true;
}

// ---- one transform of 2^L points sharded over 2^log_g ranks, on the same pass pipeline.
// With the plan's first digit R_0 and B = 2^L / R_0: rank g holds the columns [g B/G, (g+1) B/G) of
// the [R_0][B] view as a strip [R_0][B/G].  ntt_run_shard_first runs pass 0 on the strip in place (its
// inter-pass twiddle w_N^(k_0 b) IS the four-step twiddle).  After the exchange -- rank h receives
// rows k_0 in [h R_0/G, (h+1) R_0/G) of every strip and lays them out as [R_0/G][B] -- the remaining
// passes are literally passes 1.. of an (N/G)-point transform whose first digit is R_0/G
// (ntt_run_shard_rest): sub-problem sizes, twiddles and digit order are those of the global plan, and
// the last pass leaves X[k_0 + R_0 * rest] at rest * (R_0/G) + (k_0 - h R_0/G): natural-order runs
// of R_0/G outputs.  Forward transforms take any coset offset; inverse ones offset 1 (constant scale).
template <class Launcher> inline bool ntt_run_shard_first(Launcher &ln, const NttRequest &rq, uint32_t log_g, uint32_t rank) {
    if (!ntt_shard_ok(rq.L, log_g)) return false;
    const NttPlan pl = ntt_make_plan(rq.L, 1);
    const uint64_t n = 1ull << rq.L;
    PassArgs a;
    memset(&a, 0, sizeof a);
    a.in = rq.in; a.out = rq.out;
    a.in_stride = a.out_stride = n >> log_g;
    a.F = rq.F; a.T = rq.T; a.S = rq.S;
    a.L = rq.L; a.Sp = 0; a.n_in = (uint32_t)n;
    a.flags = NTT_FIRST | (rq.pre_scale ? NTT_PRE_SCALE : 0);
    a.d0_log = (uint32_t)pl.logr[0];
    a.n_tiles = (uint32_t)((n >> log_g) >> (pl.logr[0] + pl.logw[0]));
    a.batch = 1;
    const uint32_t blog = rq.L - (uint32_t)pl.logr[0];
    a.shard_log = log_g;
    a.b_off = rank << (blog - log_g);
    const uint64_t pre_step = (uint64_t)((1u << (pl.logr[0] + pl.logw[0] - 4)) >> pl.logw[0]) << blog;
    const uint32_t pr = host_powmod(rq.q_plain, pre_step, rq.F.p);
    a.pre_ratio_m = (uint32_t)(((uint64_t)pr << 32) % rq.F.p);
    a.zlog = 0;
    ln.pass(pl.logr[0], pl.logw[0], false, a, 1);
    return true;
}

// in: [R_0/G][B] (modified), out: the rank's 2^(L - log_g) outputs in the layout above
template <class Launcher> inline bool ntt_run_shard_rest(Launcher &ln, const NttRequest &rq, uint32_t log_g) {
    if (!ntt_shard_ok(rq.L, log_g)) return false;
    const NttPlan pl = ntt_make_plan(rq.L, 1);
    const uint32_t Ll = rq.L - log_g;
    const uint64_t nl = 1ull << Ll;
    uint32_t consumed = (uint32_t)pl.logr[0] - log_g;
    for (int p = 1; p < pl.np; p++) {
        const bool last = p == pl.np - 1;
        PassArgs a;
        memset(&a, 0, sizeof a);
        a.in = rq.scratch;
        a.out = last ? rq.out : rq.scratch;
        a.in_stride = a.out_stride = nl;
        a.F = rq.F; a.T = rq.T; a.S = rq.S;
        a.L = Ll; a.Sp = consumed; a.n_in = (uint32_t)nl;
        a.flags = (last && rq.post_scale ? NTT_POST_SCALE : 0) | (last && rq.last_direct ? NTT_LAST_DIRECT : 0);
        a.d0_log = (uint32_t)pl.logr[0] - log_g;
        a.n_mid = (uint32_t)(pl.np - 2);
        for (int d = 0; d < pl.np - 2; d++) a.mid_log[d] = (uint32_t)pl.logr[1 + d];
        a.n_tiles = (uint32_t)(nl >> (pl.logr[p] + pl.logw[p]));
        a.batch = 1;
        a.pre_ratio_m = a.post_ratio_m = a.post_bi_ratio_m = rq.F.r1;   // scale sequences of a sharded transform are constant (q = 1)
        ln.pass(pl.logr[p], pl.logw[p], last, a, 1);
        consumed += (uint32_t)pl.logr[p];
    }
    return true;
}
