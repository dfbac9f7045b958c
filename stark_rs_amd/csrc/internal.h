// internal.h -- context and helpers shared by the translation units of libstarkmi.so.
#pragma once
#include <string.h>
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/stark_mi.h"
#include "tables.h"

struct ScaleEntry {
    uint32_t c, q, L;
    uint32_t *lo, *hi;
    uint64_t epoch;   // smi_ctx::scale_epoch of the last ScaleScope that looked it up (pinned while that scope is open)
};

struct ProfRec {
    const char *name;
    hipEvent_t e0, e1;
    double bytes;
    double mixes;   // hash kernels: mix_state evaluations of the launch (smi_kernel_time::alg_mixes)
};

struct smi_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    FieldSetup fs;
    uint32_t *d_tab[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // [dir][tw10, lo, hi]
    std::vector<ScaleEntry> scale_cache;
    uint64_t scale_epoch = 1;   // bumped when an outermost ScaleScope opens
    int scale_depth = 0;        // open ScaleScopes (they nest: stark_prove -> fri_run -> fold)
    uint32_t *d_root_tab[32] = {};   // [log m]: w_m^e, e < m, as (value, Shoup quotient) pairs (ctx_root_table)
    uint32_t *scratch = nullptr;   // NTT inter-pass buffer
    size_t scratch_elems = 0;
    void *tmp[4] = {nullptr, nullptr, nullptr, nullptr};  // staging buffers of the host-buffer entry points
    size_t tmp_bytes[4] = {0, 0, 0, 0};
    int *d_flag = nullptr;         // non-canonical input flag
    void *pin[2] = {nullptr, nullptr};          // pinned chunks of the large host <-> device transfers
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    void *pin_out = nullptr;       // pinned landing buffer of a prove's results (proof bytes, challenges, indices, roots)
    size_t pin_out_bytes = 0;
    // a small device buffer that rides back with the next fri_run's single copy-back (the column roots of
    // smi_dev_stark_prove): set by the caller, consumed and cleared by fri_run
    const void *ride_src = nullptr;
    size_t ride_bytes = 0;
    void *ride_dst = nullptr;
    std::string err;
    bool prof_on = false;
    char prof_only[56] = {0};   // smi_ctx_profile_only: bracket only launches whose name contains this (empty: all)
    bool copy_probe = false;       // smi_ctx_copy_probe: NTT passes launch their copy-only twins
    // three-pass transforms: the second pass applies the first pass's inter-pass twiddle as it loads (ntt_core.h)
    bool ntt_share_cols = !(getenv("SMI_NTT_SHARE_COLS") && atoi(getenv("SMI_NTT_SHARE_COLS")) == 0);   // tuning knob, default on
    bool ntt_twin_regs = !(getenv("SMI_NTT_TWIN_REGS") && atoi(getenv("SMI_NTT_TWIN_REGS")) == 0);   // tuning knob, default on: deferred twiddles held as per-thread input multipliers
    bool ntt_last_direct = !(getenv("SMI_NTT_LAST_DIRECT") && atoi(getenv("SMI_NTT_LAST_DIRECT")) == 0);   // NttRequest::last_direct; tuning knob, default on
    int ntt_defer_tw = getenv("SMI_NTT_DEFER_TW") ? (atoi(getenv("SMI_NTT_DEFER_TW")) != 0) : 2;   // NttRequest::defer_tw; tuning knob, default 2 (auto)
    bool lde_two_pass = getenv("SMI_LDE_TWO_PASS") && atoi(getenv("SMI_LDE_TWO_PASS"));   // smi_ctx_lde_two_pass
    std::vector<ProfRec> prof;
    // bump arena for the per-prove device buffers (trees, folded codewords, proof bytes):
    // steady state does no hipMalloc/hipFree.  Overflow allocations are tracked and the
    // arena is regrown to the high-water mark at the next reset.
    uint8_t *arena = nullptr;
    size_t arena_size = 0, arena_used = 0, arena_want = 0;
    std::vector<void *> arena_overflow;
};
int arena_reset(smi_ctx *ctx);                       // syncs the stream if memory has to move
void *arena_alloc(smi_ctx *ctx, size_t bytes);       // nullptr on OOM

// HIP-event bracket around one kernel launch (no-ops unless profiling is enabled)
struct ProfScope {
    smi_ctx *ctx;
    ProfRec r;
    bool on;
    ProfScope(smi_ctx *c, const char *name, double bytes, double mixes = 0.0) : ctx(c), on(c->prof_on) {
        if (on && c->prof_only[0] && !strstr(name, c->prof_only)) on = false;
        if (!on) return;
        r.name = name;
        r.bytes = bytes;
        r.mixes = mixes;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.e0, ctx->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.e1, ctx->stream);
        ctx->prof.push_back(r);
    }
};

// Every public entry point runs on the context's device whatever the caller's current device is
// (a process may hold contexts on several GPUs, or torch may have selected another one), and
// leaves the caller's selection as it found it.
struct DeviceGuard {
    int prev = -1;
    bool moved = false;
    explicit DeviceGuard(const smi_ctx *ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) moved = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (moved) (void)hipSetDevice(prev);
    }
};

struct smi_tree {
    smi_ctx *ctx;
    uint8_t *d_nodes;   // (2n-1) x 32 bytes, level 0 first
    size_t n;
    bool owns;
};

int smi_hip_fail(smi_ctx *ctx, hipError_t e, const char *what);
#define HIP_TRY(ctx, call)                                             \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return smi_hip_fail((ctx), e__, #call); \
    } while (0)
#define SMI_TRY(call)            \
    do {                         \
        int rc__ = (call);       \
        if (rc__ != SMI_OK) return rc__; \
    } while (0)

int smi_fail(smi_ctx *ctx, int code, const char *msg);

// grows (never shrinks) a context-owned staging buffer
int ctx_tmp(smi_ctx *ctx, int slot, size_t bytes, void **out);
int ctx_scratch(smi_ctx *ctx, size_t elems, uint32_t **out);
// Pinned host memory for the results a prove copies back (grows, never shrinks): a device-to-host copy into pageable
// memory is staged and blocks per call (r03 trace: 6 copies, 105 us of idle at the end of a prove); into pinned
// memory the copies queue behind the last kernel and one synchronisation waits for all of them.
int ctx_pin_out(smi_ctx *ctx, size_t bytes, uint8_t **out);
NttTables ctx_tables(const smi_ctx *ctx, int inverse);
// device tables of c * q^i, i < 2^L (cached per (c,q,L))
// device table of w_m^e, e < m = 2^log_m (forward root), as Tw2 pairs; cached per log_m (log_m <= 17)
int ctx_root_table(smi_ctx *ctx, uint32_t log_m, const Tw2 **out);
int ctx_scale_tables(smi_ctx *ctx, uint32_t c_plain, uint32_t q_plain, uint32_t L, ScaleTables *out);
#define SMI_SCALE_CACHE_MAX 96   // soft cap: entries pinned by an open ScaleScope are never evicted, the cache grows instead
int ctx_scale_reserve(smi_ctx *ctx, size_t n);   // room for n more entries (evicts unpinned ones only)
// Pins every scale table looked up while it is open: ctx_scale_tables hands out raw device pointers, and a caller
// that holds one (or collects several, like the fused FRI tail) across another lookup must not see it freed by
// that lookup's eviction.  Open one in every function that takes tables and enqueues the launches that read
// them; an eviction drains the stream before it frees, so a table is safe once its launch is enqueued.
struct ScaleScope {
    smi_ctx *c;
    explicit ScaleScope(smi_ctx *ctx) : c(ctx) {
        if (c->scale_depth++ == 0) c->scale_epoch++;
    }
    ~ScaleScope() { c->scale_depth--; }
    ScaleScope(const ScaleScope &) = delete;
    ScaleScope &operator=(const ScaleScope &) = delete;
};

// field helpers on the host (plain form)
inline uint32_t h_mul(const smi_ctx *c, uint32_t a, uint32_t b) { return host_mulmod(a, b, c->fs.F.p); }
inline uint32_t h_pow(const smi_ctx *c, uint32_t a, uint64_t e) { return host_powmod(a, e, c->fs.F.p); }
inline uint32_t h_inv(const smi_ctx *c, uint32_t a) { return host_powmod(a, c->fs.F.p - 2, c->fs.F.p); }
inline uint32_t h_root(const smi_ctx *c, uint32_t log_n) {  // primitive 2^log_n-th root (forward)
    return host_powmod(c->fs.wmax[0], 1ull << (c->fs.K - log_n), c->fs.F.p);
}

// One launch for the last rounds of Fri::commit (hash.hip, fri_tail_kernel): round k hashes and commits
// cw (len elements), runs the Fiat-Shamir round of its root and, unless next == nullptr (the last
// round), folds into next with the round's x^-1 table S.
#define SMI_FRI_TAIL_MAX_ROUNDS 12
#define SMI_FRI_TAIL_MAX_LEN 2048   // = SMI_TOP_MAX of hash.hip
struct FriTailRound {
    const uint32_t *cw;
    uint32_t *next;
    uint8_t *nodes;
    uint8_t *proof_slot;
    uint64_t *alpha_out;   // nullptr on the last round
    ScaleTables S;
    uint32_t len;
};
struct FriTailArgs {
    FriTailRound r[SMI_FRI_TAIL_MAX_ROUNDS];
    uint32_t n_rounds;
    uint32_t *fs_words;
    Fp F;
    uint32_t inv2_m;
    // optional: the fold that produces r[0].cw (from the halves pre_lo / pre_hi of the round before, with its challenge
    // and x^-1 table) runs at the head of the launch instead of in a fold launch of its own; pre_lo == nullptr: it does not
    const uint32_t *pre_lo, *pre_hi;
    const uint64_t *pre_alpha;
    ScaleTables pre_S;
};
int launch_fri_tail(smi_ctx *ctx, const FriTailArgs &a);
uint64_t fri_tail_len();   // codewords of at most this many elements finish in the fused tail (SMI_FRI_TAIL, default 512; fri.hip)

// Where the leaves of a tree come from when they are not simply read (hash.hip, merkle_sub_kernel's LEAF_* kinds): the
// kernel computes the codeword element, stores it to cw_out (the query phase and the next fold read it) and hashes it.
// The hash kernels are bound by integer issue and leave HBM idle, so the fold's / the combination's reads and writes ride
// along for the price of their few arithmetic instructions -- one launch and one pass over the codeword less per round.
//   LEAF_FOLD   : element i = Fri::fold_codeword(lo[i], hi[i]) (src/fri.rs:57-91; fri_core.h fold_element, the function the
//                 stand-alone fri_fold_kernel runs) with the round's challenge read from device memory;
//   LEAF_COMBINE: element i = sum_c (weights[c] mod p) * cols[c * stride + i] (combine_columns_kernel's sum, same order).
enum { LEAF_LOAD = 0, LEAF_FOLD = 1, LEAF_COMBINE = 2 };
#define SMI_LEAF_COMBINE_MAX 8
struct LeafSrc {
    int kind;
    uint32_t *cw_out;
    Fp F;
    // LEAF_FOLD
    const uint32_t *lo, *hi;
    const uint64_t *alpha;
    ScaleTables S;
    uint32_t inv2_m;
    // LEAF_COMBINE
    const uint32_t *cols;
    size_t stride;
    uint32_t n_cols;
    const uint32_t *weights_m;   // (weight mod p) in Montgomery form, on the device (fs_weights_kernel)
};
// true when a tree of n single-element leaves starts with the thread-per-four-leaves kernel that can take a LeafSrc
bool merkle_fuses_leaf_source(size_t n);
// true when a tree of n single-element leaves starts with the chunk kernel, which can take a LEAF_FOLD source (no other
// kind; no alignment demands: it reads and writes single elements)
bool merkle_chunks_fold(size_t n);
int launch_merkle_src_fs(smi_ctx *ctx, const LeafSrc &src, size_t n, uint8_t *d_nodes, uint32_t *fs_words, uint8_t *proof_slot,
                         uint64_t *alpha_out, bool *done);

// launches (defined in the .hip files)
int launch_geom_table(smi_ctx *ctx, const GeomSpec &s, uint32_t *d_out);
int launch_narrow(smi_ctx *ctx, const uint64_t *d_in, uint32_t *d_out, size_t n, int reduce);
int launch_widen(smi_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, size_t n);
int dev_ntt(smi_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint32_t log_n, size_t n_in, uint32_t batch,
            size_t in_stride, size_t out_stride, int inverse, uint64_t offset, uint64_t post_scale);
int dev_lde2(smi_ctx *ctx, const uint32_t *d_coef, uint32_t *d_out, uint32_t log_n, uint32_t log_blowup, uint32_t batch,
             size_t coef_stride, size_t out_stride);
int dev_ntt_shard_first(smi_ctx *ctx, uint32_t *d_strip, uint32_t log_n, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset);
int dev_ntt_shard_rest(smi_ctx *ctx, uint32_t *d_rows, uint32_t *d_out, uint32_t log_n, uint32_t log_g, int inverse);
int check_flag(smi_ctx *ctx);  // syncs; SMI_ERR_NON_CANONICAL if a narrow kernel saw a value >= p
// caller's (pageable) u64 buffers <-> device u32 residues; synchronous on return
int host_to_dev_u32(smi_ctx *ctx, const uint64_t *host, size_t n, uint32_t *d_out, int reduce);
int dev_u32_to_host(smi_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *host);
