// api.hip -- context management and the extern "C" surface declared in include/stark_mi.h.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "hash_core.h"
#include "internal.h"
#include "lde_core.h"

int launch_leaf_hash(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_digests);
int launch_combine(smi_ctx *ctx, const uint8_t *d_in, size_t n_pairs, uint8_t *d_out);
int launch_hash_bytes(smi_ctx *ctx, const uint8_t *d_msg, size_t len, uint32_t *d_out);
int launch_merkle(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes);
int launch_verify_paths(smi_ctx *ctx, const uint8_t *d_leaves, const uint64_t *d_idx, const uint8_t *d_paths, size_t k, uint32_t depth,
                        const uint8_t *d_root, uint8_t *d_ok);

// ------------------------------------------------------------------------- errors
const char *smi_status_string(int status) {
    switch (status) {
    case SMI_OK: return "ok";
    case SMI_ERR_NO_INVERSE: return "no inverse";
    case SMI_ERR_DIV_BY_ZERO: return "no division by zero";
    case SMI_ERR_POLY_DIV_BY_ZERO: return "No division by zero";
    case SMI_ERR_NOT_POW2: return "n must be a power of two";
    case SMI_ERR_ROOT_TOO_LARGE: return "n > 2^23 not supported by this modulus";
    case SMI_ERR_EMPTY_LEAVES: return "Cannot create tree from empty leaves";
    case SMI_ERR_LEAVES_NOT_POW2: return "Number of leaves must be power of 2";
    case SMI_ERR_INDEX_OOB: return "Index out of bounds";
    case SMI_ERR_DOMAIN_NOT_POW2: return "Domain length must be power of 2";
    case SMI_ERR_EXPANSION_NOT_POW2: return "Expansion factor must be power of 2";
    case SMI_ERR_EXPANSION_TOO_SMALL: return "Expansion factor must be at least 4";
    case SMI_ERR_CODEWORD_LEN: return "initial codeword length does not match domain length";
    case SMI_ERR_SAMPLE_ENTROPY: return "not enough entropy in indices wrt last codeword";
    case SMI_ERR_SAMPLE_TOO_MANY: return "cannot sample more indices than available in last codeword";
    case SMI_ERR_LEN_MISMATCH: return "assertion failed: domain.len() == values.len()";
    case SMI_ERR_EMPTY_DOMAIN: return "assertion failed: domain.len() > 0";
    case SMI_ERR_WRONG_FIELD: return "assertion failed: self.p == 998244353";
    case SMI_ERR_NO_ROUNDS: return "No FRI roots extracted";
    case SMI_ERR_BAD_ARG: return "bad argument";
    case SMI_ERR_NON_CANONICAL: return "field value not canonical (>= p)";
    case SMI_ERR_UNSUPPORTED_PRIME: return "unsupported modulus for this size";
    case SMI_ERR_NOT_GEOMETRIC: return "domain is not offset*omega^k";
    case SMI_ERR_COLUMNS_NOT_BOUND: return "proof has no column openings: it does not bind the column roots";
    case SMI_ERR_HIP: return "HIP runtime error";
    case SMI_ERR_NO_DEVICE: return "no usable HIP device";
    case SMI_ERR_OOM: return "out of memory";
    case SMI_ERR_RCCL: return "RCCL / collective error";
    default: return "unknown status";
    }
}
const char *smi_version(void) { return "stark-mi 0.1 (gfx950)"; }
const char *smi_last_error(const smi_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int smi_fail(smi_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg ? msg : smi_status_string(code);
    return code;
}
int smi_hip_fail(smi_ctx *ctx, hipError_t e, const char *what) {
    if (ctx) ctx->err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SMI_ERR_OOM : SMI_ERR_HIP;
}

// ------------------------------------------------------------------------- context
int smi_ctx_create(uint64_t p, uint64_t g, int device, smi_ctx **out) {
    if (!out) return SMI_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return SMI_ERR_NO_DEVICE;
    smi_ctx *ctx = new smi_ctx();
    if (!field_setup(p, g, &ctx->fs)) {
        delete ctx;
        return SMI_ERR_UNSUPPORTED_PRIME;
    }
    ctx->device = device;
    int rc = SMI_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = SMI_ERR_NO_DEVICE; break; }
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->num_cus = cus;
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { rc = SMI_ERR_HIP; break; }
        ctx->own_stream = true;
        if (hipMalloc((void **)&ctx->d_flag, sizeof(int)) != hipSuccess) { rc = SMI_ERR_OOM; break; }
        if (hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->stream) != hipSuccess) { rc = SMI_ERR_HIP; break; }
        for (int dir = 0; dir < 2 && rc == SMI_OK; dir++) {
            GeomSpec sp[3];
            ntt_table_specs(ctx->fs, dir, sp);
            for (int k = 0; k < 3 && rc == SMI_OK; k++) {
                if (hipMalloc((void **)&ctx->d_tab[dir][k], ((size_t)sp[k].count * 4) << sp[k].pair) != hipSuccess) { rc = SMI_ERR_OOM; break; }
                rc = launch_geom_table(ctx, sp[k], ctx->d_tab[dir][k]);
            }
        }
        if (rc != SMI_OK) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = SMI_ERR_HIP; break; }
    } while (0);
    if (rc != SMI_OK) {
        smi_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return SMI_OK;
}

void smi_ctx_destroy(smi_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard dg__(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int d = 0; d < 2; d++)
        for (int k = 0; k < 3; k++) (void)hipFree(ctx->d_tab[d][k]);
    for (ScaleEntry &e : ctx->scale_cache) {
        (void)hipFree(e.lo);
        (void)hipFree(e.hi);
    }
    for (uint32_t *t : ctx->d_root_tab) (void)hipFree(t);
    (void)hipFree(ctx->scratch);
    (void)hipFree(ctx->arena);
    for (void *q : ctx->arena_overflow) (void)hipFree(q);
    for (int i = 0; i < 4; i++) (void)hipFree(ctx->tmp[i]);
    for (int i = 0; i < 2; i++) {
        if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]);
        if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
    }
    if (ctx->pin_out) (void)hipHostFree(ctx->pin_out);
    (void)hipFree(ctx->d_flag);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int smi_ctx_set_stream(smi_ctx *ctx, void *hip_stream) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return SMI_OK;
}
int smi_ctx_sync(smi_ctx *ctx) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}
int smi_ctx_profile(smi_ctx *ctx, int enable) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    ctx->prof_on = enable != 0;
    return SMI_OK;
}
int smi_ctx_profile_only(smi_ctx *ctx, const char *name_part) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    if (name_part && strlen(name_part) >= sizeof ctx->prof_only) return smi_fail(ctx, SMI_ERR_BAD_ARG, "profile filter too long");
    memset(ctx->prof_only, 0, sizeof ctx->prof_only);
    if (name_part) memcpy(ctx->prof_only, name_part, strlen(name_part));
    return SMI_OK;
}
int smi_ctx_lde_two_pass(smi_ctx *ctx, int enable) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    ctx->lde_two_pass = enable != 0;
    return SMI_OK;
}
int smi_ctx_copy_probe(smi_ctx *ctx, int enable) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    ctx->copy_probe = enable != 0;
    return SMI_OK;
}
int smi_ctx_profile_read(smi_ctx *ctx, smi_kernel_time *out, size_t cap, size_t *n) {
    if (!ctx || !n || (cap && !out)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    size_t cnt = 0;
    for (ProfRec &r : ctx->prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
        size_t k = 0;
        for (; k < cnt; k++)
            if (!strncmp(out[k].name, r.name, sizeof out[k].name - 1)) break;
        if (k == cnt) {
            if (cnt == cap) continue;
            memset(&out[k], 0, sizeof out[k]);
            strncpy(out[k].name, r.name, sizeof out[k].name - 1);
            cnt++;
        }
        out[k].launches++;
        out[k].total_ms += ms;
        out[k].alg_bytes += r.bytes;
        out[k].alg_mixes += r.mixes;
    }
    ctx->prof.clear();
    *n = cnt;
    return SMI_OK;
}
uint64_t smi_ctx_modulus(const smi_ctx *ctx) { return ctx ? ctx->fs.F.p : 0; }
uint32_t smi_ctx_two_adicity(const smi_ctx *ctx) { return ctx ? ctx->fs.K : 0; }

int ctx_pin_out(smi_ctx *ctx, size_t bytes, uint8_t **out) {
    if (bytes > ctx->pin_out_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // a copy into the old buffer may still be in flight
        if (ctx->pin_out) (void)hipHostFree(ctx->pin_out);
        ctx->pin_out = nullptr;
        ctx->pin_out_bytes = 0;
        const size_t want = bytes < ((size_t)2 << 20) ? ((size_t)2 << 20) : bytes + bytes / 4;
        if (hipHostMalloc(&ctx->pin_out, want, hipHostMallocDefault) != hipSuccess) {
            ctx->pin_out = nullptr;
            return smi_fail(ctx, SMI_ERR_OOM, "hipHostMalloc (result buffer)");
        }
        ctx->pin_out_bytes = want;
    }
    *out = (uint8_t *)ctx->pin_out;
    return SMI_OK;
}
int ctx_tmp(smi_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes > ctx->tmp_bytes[slot]) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->tmp[slot]);
        ctx->tmp[slot] = nullptr;
        ctx->tmp_bytes[slot] = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&ctx->tmp[slot], want) != hipSuccess) return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc staging buffer");
        ctx->tmp_bytes[slot] = want;
    }
    *out = ctx->tmp[slot];
    return SMI_OK;
}
int arena_reset(smi_ctx *ctx) {
    if (!ctx->arena_overflow.empty() || ctx->arena_want > ctx->arena_size) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (void *q : ctx->arena_overflow) (void)hipFree(q);
        ctx->arena_overflow.clear();
        if (ctx->arena_want > ctx->arena_size) {
            (void)hipFree(ctx->arena);
            ctx->arena = nullptr;
            ctx->arena_size = 0;
            const size_t want = ctx->arena_want + ctx->arena_want / 16 + (1u << 20);
            if (hipMalloc((void **)&ctx->arena, want) == hipSuccess) ctx->arena_size = want;
            else (void)hipGetLastError();
        }
    }
    ctx->arena_used = 0;
    ctx->arena_want = 0;
    return SMI_OK;
}
void *arena_alloc(smi_ctx *ctx, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    ctx->arena_want += bytes;
    if (ctx->arena_used + bytes <= ctx->arena_size) {
        void *q = ctx->arena + ctx->arena_used;
        ctx->arena_used += bytes;
        return q;
    }
    void *q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 256) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    ctx->arena_overflow.push_back(q);
    return q;
}
int ctx_scratch(smi_ctx *ctx, size_t elems, uint32_t **out) {
    if (elems > ctx->scratch_elems) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_elems = 0;
        if (hipMalloc((void **)&ctx->scratch, elems * 4) != hipSuccess) return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc NTT scratch");
        ctx->scratch_elems = elems;
    }
    *out = ctx->scratch;
    return SMI_OK;
}
NttTables ctx_tables(const smi_ctx *ctx, int inverse) {
    const int d = inverse ? 1 : 0;
    return NttTables{(const Tw2 *)ctx->d_tab[d][0], ctx->d_tab[d][1], ctx->d_tab[d][2], ctx->fs.K, ntt_table_h(ctx->fs.K)};
}
int ctx_root_table(smi_ctx *ctx, uint32_t log_m, const Tw2 **out) {
    if (log_m > 17 || log_m > ctx->fs.K) return smi_fail(ctx, SMI_ERR_BAD_ARG, "root table too large");
    if (!ctx->d_root_tab[log_m]) {
        const Fp &F = ctx->fs.F;
        const uint32_t w = h_root(ctx, log_m);
        const GeomSpec sp{F.r1, (uint32_t)(((uint64_t)w << 32) % F.p), 1, 1u << log_m, 1};
        uint32_t *d = nullptr;
        if (hipMalloc((void **)&d, ((size_t)8) << log_m) != hipSuccess) return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc root table");
        const int rc = launch_geom_table(ctx, sp, d);
        if (rc != SMI_OK) {
            (void)hipFree(d);
            return rc;
        }
        ctx->d_root_tab[log_m] = d;
    }
    *out = (const Tw2 *)ctx->d_root_tab[log_m];
    return SMI_OK;
}
// Makes room for `n` new entries: evicts the oldest unpinned entries (internal.h, ScaleScope) after draining the
// stream -- in-flight kernels may still read them.  Entries looked up under the open scope stay; if everything is
// pinned the cache simply grows past its soft cap.
int ctx_scale_reserve(smi_ctx *ctx, size_t n) {
    if (ctx->scale_cache.size() + n < SMI_SCALE_CACHE_MAX) return SMI_OK;
    auto pinned = [&](const ScaleEntry &e) { return ctx->scale_depth > 0 && e.epoch == ctx->scale_epoch; };
    size_t want = ctx->scale_cache.size() / 2 > n ? ctx->scale_cache.size() / 2 : ctx->scale_cache.size();
    bool any = false;
    for (const ScaleEntry &e : ctx->scale_cache) any |= !pinned(e);
    if (!any) return SMI_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<ScaleEntry> keep;
    keep.reserve(ctx->scale_cache.size());
    for (const ScaleEntry &e : ctx->scale_cache) {   // oldest first
        if (want && !pinned(e)) {
            (void)hipFree(e.lo);
            (void)hipFree(e.hi);
            want--;
        } else {
            keep.push_back(e);
        }
    }
    ctx->scale_cache.swap(keep);
    return SMI_OK;
}
int ctx_scale_tables(smi_ctx *ctx, uint32_t c_plain, uint32_t q_plain, uint32_t L, ScaleTables *out) {
    for (ScaleEntry &e : ctx->scale_cache)
        if (e.c == c_plain && e.q == q_plain && e.L == L) {
            e.epoch = ctx->scale_epoch;
            *out = ScaleTables{e.lo, e.hi, scale_table_h(L)};
            return SMI_OK;
        }
    SMI_TRY(ctx_scale_reserve(ctx, 1));
    GeomSpec sp[2];
    scale_table_specs(ctx->fs.F, c_plain, q_plain, L, sp);
    ScaleEntry e{c_plain, q_plain, L, nullptr, nullptr, ctx->scale_epoch};
    if (hipMalloc((void **)&e.lo, (size_t)sp[0].count * 4) != hipSuccess) return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc scale table");
    if (hipMalloc((void **)&e.hi, (size_t)sp[1].count * 4) != hipSuccess) {
        (void)hipFree(e.lo);
        return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc scale table");
    }
    int rc = launch_geom_table(ctx, sp[0], e.lo);
    if (rc == SMI_OK) rc = launch_geom_table(ctx, sp[1], e.hi);
    if (rc != SMI_OK) {
        (void)hipFree(e.lo);
        (void)hipFree(e.hi);
        return rc;
    }
    ctx->scale_cache.push_back(e);
    *out = ScaleTables{e.lo, e.hi, scale_table_h(L)};
    return SMI_OK;
}
int check_flag(smi_ctx *ctx) {
    int flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->stream));
        return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "input field value >= p");
    }
    return SMI_OK;
}

// ---- large transfers between the caller's buffers and the device.  Field values cross the ABI as
// u64 (the reference's wire width, src/stream.rs:45) in ordinary pageable memory, but a residue is 32
// bits: only the u32 halves cross PCIe, by DMA into / out of two pinned chunks, while host threads
// widen (or narrow and range-check) the previous chunk against the caller's buffer.  Small transfers
// keep the plain path (one staged copy + a device kernel).
static constexpr size_t SMI_PIN_CHUNK = (size_t)8 << 20;   // elements per pinned chunk (32 MiB)
static constexpr size_t SMI_PIN_MIN = (size_t)1 << 20;     // below this many elements: plain path

static int ctx_pinned(smi_ctx *ctx) {
    for (int i = 0; i < 2; i++) {
        if (!ctx->pin[i] && hipHostMalloc(&ctx->pin[i], SMI_PIN_CHUNK * 4, hipHostMallocDefault) != hipSuccess) {
            ctx->pin[i] = nullptr;
            return smi_fail(ctx, SMI_ERR_OOM, "hipHostMalloc");
        }
        if (!ctx->pin_ev[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
    }
    return SMI_OK;
}
// Conversion threads: a small persistent pool (creating threads per chunk cost more than the chunk's
// conversion: 2^22 x 4 LDE into a reused buffer 16.9 -> 14 ms).  SMI_HOST_THREADS overrides the default
// of up to 8 (16 measured no better on the 16-core share of a GPU box); calls are serialized.
namespace {
class HostPool {
    std::vector<std::thread> th_;
    std::mutex m_, run_m_;
    std::condition_variable work_, done_;
    std::function<void(size_t, size_t)> fn_;
    size_t n_ = 0, per_ = 0;
    unsigned gen_ = 0, pending_ = 0;
    bool stop_ = false;

    void worker(unsigned i) {
        unsigned seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            work_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            const size_t lo = (size_t)i * per_, hi = lo + per_ < n_ ? lo + per_ : n_;
            lk.unlock();
            if (lo < hi) fn_(lo, hi);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }

public:
    explicit HostPool(unsigned t) {
        for (unsigned i = 1; i < t; i++) th_.emplace_back([this, i] { worker(i); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        work_.notify_all();
        for (std::thread &x : th_) x.join();
    }
    unsigned size() const { return (unsigned)th_.size() + 1; }
    void run(size_t n, std::function<void(size_t, size_t)> fn) {   // fn(lo, hi) over [0, n), the caller takes the first slice
        std::lock_guard<std::mutex> serial(run_m_);
        const unsigned t = size();
        const size_t per = (n + t - 1) / t;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = fn;
            n_ = n;
            per_ = per;
            pending_ = t - 1;
            gen_++;
        }
        work_.notify_all();
        fn(0, per < n ? per : n);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }
};
}  // namespace

template <class Fn> static void host_parallel(size_t n, Fn fn) {
    if (n < ((size_t)1 << 18)) {
        fn(0, n);
        return;
    }
    static HostPool pool([] {
        const char *e = getenv("SMI_HOST_THREADS");
        const int v = e ? atoi(e) : 0;
        unsigned hw = std::thread::hardware_concurrency();
        hw = hw < 1 ? 1 : hw;
        const unsigned cap = (unsigned)(v >= 1 && v <= 64 ? v : 8);
        return hw < cap ? hw : cap;
    }());
    pool.run(n, fn);
}

int dev_u32_to_host(smi_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *host) {
    if (!n) return SMI_OK;
    if (n < SMI_PIN_MIN) {
        void *stage;
        SMI_TRY(ctx_tmp(ctx, 0, n * 8, &stage));
        SMI_TRY(launch_widen(ctx, d_in, (uint64_t *)stage, n));
        HIP_TRY(ctx, hipMemcpyAsync(host, stage, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return SMI_OK;
    }
    SMI_TRY(ctx_pinned(ctx));
    const size_t nch = (n + SMI_PIN_CHUNK - 1) / SMI_PIN_CHUNK;
    for (size_t c = 0; c <= nch; c++) {
        if (c < nch) {   // chunk c leaves the device while chunk c-1 is widened below
            const size_t off = c * SMI_PIN_CHUNK, cnt = n - off < SMI_PIN_CHUNK ? n - off : SMI_PIN_CHUNK;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->pin[c & 1], d_in + off, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->pin_ev[c & 1], ctx->stream));
        }
        if (c > 0) {
            const size_t off = (c - 1) * SMI_PIN_CHUNK, cnt = n - off < SMI_PIN_CHUNK ? n - off : SMI_PIN_CHUNK;
            HIP_TRY(ctx, hipEventSynchronize(ctx->pin_ev[(c - 1) & 1]));
            const uint32_t *src = (const uint32_t *)ctx->pin[(c - 1) & 1];
            uint64_t *dst = host + off;
            host_parallel(cnt, [=](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; i++) dst[i] = src[i];
            });
        }
    }
    return SMI_OK;
}

int host_to_dev_u32(smi_ctx *ctx, const uint64_t *host, size_t n, uint32_t *d_out, int reduce) {
    if (!n) return SMI_OK;
    if (n < SMI_PIN_MIN) {
        void *stage;
        SMI_TRY(ctx_tmp(ctx, 0, n * 8, &stage));
        HIP_TRY(ctx, hipMemcpyAsync(stage, host, n * 8, hipMemcpyHostToDevice, ctx->stream));
        SMI_TRY(launch_narrow(ctx, (const uint64_t *)stage, d_out, n, reduce));
        return check_flag(ctx);
    }
    SMI_TRY(ctx_pinned(ctx));
    const uint64_t p = ctx->fs.F.p;
    const size_t nch = (n + SMI_PIN_CHUNK - 1) / SMI_PIN_CHUNK;
    std::vector<int> bad(nch, 0);
    for (size_t c = 0; c < nch; c++) {
        const size_t off = c * SMI_PIN_CHUNK, cnt = n - off < SMI_PIN_CHUNK ? n - off : SMI_PIN_CHUNK;
        if (c >= 2) HIP_TRY(ctx, hipEventSynchronize(ctx->pin_ev[c & 1]));   // chunk c-2 has left this buffer
        uint32_t *dst = (uint32_t *)ctx->pin[c & 1];
        const uint64_t *src = host + off;
        int *flag = &bad[c];
        host_parallel(cnt, [=](size_t lo, size_t hi) {
            uint64_t over = 0;
            if (reduce) {
                for (size_t i = lo; i < hi; i++) dst[i] = (uint32_t)(src[i] % p);
            } else {
                for (size_t i = lo; i < hi; i++) {
                    over |= (uint64_t)(src[i] >= p);
                    dst[i] = (uint32_t)src[i];
                }
            }
            if (over) __atomic_store_n(flag, 1, __ATOMIC_RELAXED);
        });
        HIP_TRY(ctx, hipMemcpyAsync(d_out + off, dst, cnt * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->pin_ev[c & 1], ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t c = 0; c < nch; c++)
        if (bad[c]) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "input field value >= p");
    return SMI_OK;
}

// ------------------------------------------------------------------------- field scalars
static bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }
static uint32_t ilog2(uint64_t n) {
    uint32_t l = 0;
    while ((n >> l) > 1) l++;
    return l;
}

int smi_prim_nth_root(const smi_ctx *ctx, uint64_t n, uint64_t *out) {
    if (!ctx || !out) return SMI_ERR_BAD_ARG;
    if (!is_pow2(n)) return SMI_ERR_NOT_POW2;                      // src/ff.rs:217
    const uint32_t p = ctx->fs.F.p;
    if (ilog2(n) > ctx->fs.K) return p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME;  // src/ff.rs:218
    *out = host_powmod(ctx->fs.g, (p - 1) / n, p);                 // g^((p-1)/n), src/ff.rs:220-222
    return SMI_OK;
}
int smi_ff_inv(const smi_ctx *ctx, uint64_t x, uint64_t *out) {
    if (!ctx || !out) return SMI_ERR_BAD_ARG;
    const uint32_t p = ctx->fs.F.p;
    if (x % p == 0) return SMI_ERR_NO_INVERSE;
    *out = h_inv(ctx, (uint32_t)(x % p));
    return SMI_OK;
}
int smi_ff_exp(const smi_ctx *ctx, uint64_t base, uint64_t e, uint64_t *out) {
    if (!ctx || !out) return SMI_ERR_BAD_ARG;
    *out = h_pow(ctx, (uint32_t)(base % ctx->fs.F.p), e);
    return SMI_OK;
}
int smi_ff_mul(const smi_ctx *ctx, uint64_t a, uint64_t b, uint64_t *out) {
    if (!ctx || !out) return SMI_ERR_BAD_ARG;
    const uint32_t p = ctx->fs.F.p;
    *out = h_mul(ctx, (uint32_t)(a % p), (uint32_t)(b % p));
    return SMI_OK;
}
int smi_domain_is_geometric(const smi_ctx *ctx, const uint64_t *domain, size_t n, uint64_t *offset) {
    if (!ctx || !domain || !n) return SMI_ERR_BAD_ARG;
    if (!is_pow2(n) || ilog2(n) > ctx->fs.K) return SMI_ERR_NOT_GEOMETRIC;
    const uint32_t p = ctx->fs.F.p, w = h_root(ctx, ilog2(n));
    if (domain[0] == 0 || domain[0] >= p) return SMI_ERR_NOT_GEOMETRIC;
    uint32_t x = (uint32_t)domain[0];
    for (size_t k = 1; k < n; k++) {
        x = host_mulmod(x, w, p);
        if (domain[k] != x) return SMI_ERR_NOT_GEOMETRIC;
    }
    if (offset) *offset = domain[0];
    return SMI_OK;
}

// ------------------------------------------------------------------------- device memory
int smi_dev_alloc(smi_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc");
    return SMI_OK;
}
int smi_dev_free(smi_ctx *ctx, void *d_ptr) {
    if (!ctx) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(d_ptr));
    return SMI_OK;
}
int smi_dev_upload_u64(smi_ctx *ctx, const uint64_t *host, size_t n, uint32_t *d_out, int reduce) {
    if (!ctx || (!host && n) || (!d_out && n)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return host_to_dev_u32(ctx, host, n, d_out, reduce);
}
int smi_dev_download_u64(smi_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *host) {
    if (!ctx || (!host && n) || (!d_in && n)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return dev_u32_to_host(ctx, d_in, n, host);
}

// ------------------------------------------------------------------------- univariate
int smi_dev_ntt(smi_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint32_t log_n, size_t n_in, uint32_t batch,
                size_t in_stride, size_t out_stride, int inverse, uint64_t offset, uint64_t post_scale) {
    if (!ctx || !d_in || !d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return dev_ntt(ctx, d_in, d_out, log_n, n_in, batch, in_stride, out_stride, inverse, offset, post_scale);
}

int smi_dev_lde(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, uint32_t log_n, uint32_t log_blowup,
                uint64_t trace_offset, uint64_t lde_offset, uint32_t *d_out) {
    if (!ctx || !d_cols || !d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    const uint32_t log_N = log_n + log_blowup;
    if (log_N > ctx->fs.K) return smi_fail(ctx, ctx->fs.F.p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME,
                                           "LDE domain exceeds the two-adicity of the modulus");
    const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
    // interpolate on trace_offset*w_n^k; the coset shift of the evaluation (Polynomial::scale by
    // lde_offset, src/univariate/mod.rs:99-113) is fused into the inverse transform's output
    // scaling, so the forward transform runs with offset 1 and reads the n coefficients from
    // the head of each output column (zero-padded to N on the fly).
    auto extend = [&](const uint32_t *cols, uint32_t w, uint32_t *out) -> int {
        SMI_TRY(dev_ntt(ctx, cols, out, log_n, n, w, n, N, 1, trace_offset, lde_offset));
        // The extension's zero padding lets it run in two passes over the outputs instead of three
        // (lde_core.h).  Measured on MI355X: both of its passes end up as close to
        // their arithmetic as to their memory time and the step is within 2 % of the generic transform's,
        // so the generic path stays the default; smi_ctx_lde_two_pass / SMI_LDE_TWO_PASS=1 select it.
        if (ctx->lde_two_pass && lde2_supported(log_n, log_blowup)) return dev_lde2(ctx, out, out, log_n, log_blowup, w, N, N);
        return dev_ntt(ctx, out, out, log_N, n, w, N, N, 0, 1, 1);
    };
    return extend(d_cols, n_cols, d_out);
}

static int host_ntt(smi_ctx *ctx, const uint64_t *in, size_t n_in, uint64_t *out, uint32_t log_n, int inverse, uint64_t offset) {
    const size_t n = (size_t)1 << log_n;
    void *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 1, (n_in ? n_in : 1) * 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, n * 4, &d_out));
    SMI_TRY(host_to_dev_u32(ctx, in, n_in, (uint32_t *)d_in, 0));
    SMI_TRY(dev_ntt(ctx, (const uint32_t *)d_in, (uint32_t *)d_out, log_n, n_in, 1, n_in, n, inverse, offset, 1));
    return dev_u32_to_host(ctx, (const uint32_t *)d_out, n, out);
}

int smi_intt(smi_ctx *ctx, const uint64_t *values, uint64_t *coeffs, uint32_t log_n, uint64_t offset) {
    if (!ctx || !values || !coeffs) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (log_n > 40) return SMI_ERR_BAD_ARG;
    if (offset % ctx->fs.F.p == 0) return smi_fail(ctx, SMI_ERR_NO_INVERSE, "no inverse");  // duplicate domain points, src/ff.rs:171
    return host_ntt(ctx, values, (size_t)1 << log_n, coeffs, log_n, 1, offset);
}
int smi_coset_ntt(smi_ctx *ctx, const uint64_t *coeffs, size_t n_coeffs, uint64_t *evals, uint32_t log_N, uint64_t offset) {
    if (!ctx || (!coeffs && n_coeffs) || !evals) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (log_N > 40 || n_coeffs > ((size_t)1 << log_N)) return SMI_ERR_BAD_ARG;
    if (offset % ctx->fs.F.p == 0) return smi_fail(ctx, SMI_ERR_BAD_ARG, "coset offset must be nonzero");
    return host_ntt(ctx, coeffs, n_coeffs, evals, log_N, 0, offset);
}

// out[i] = coeffs[i] * factor^i -- Polynomial::scale (src/univariate/mod.rs:99-113)
__global__ void scale_kernel(const uint32_t *in, uint32_t *out, size_t n, Fp F, ScaleTables S) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = mont_mul(in[i], two_level(S.lo, S.hi, S.h, (uint32_t)i, F), F);
}
int smi_poly_scale(smi_ctx *ctx, const uint64_t *coeffs, size_t n, uint64_t factor, uint64_t *out) {
    if (!ctx || (!coeffs && n) || (!out && n)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (!n) return SMI_OK;
    if (factor >= ctx->fs.F.p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "factor must be < p");
    uint32_t L = 0;
    while (((size_t)1 << L) < n) L++;
    void *stage, *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 0, n * 8, &stage));
    SMI_TRY(ctx_tmp(ctx, 1, n * 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, n * 4, &d_out));
    ScaleScope pin__(ctx);
    ScaleTables S;
    SMI_TRY(ctx_scale_tables(ctx, 1, (uint32_t)factor, L, &S));
    HIP_TRY(ctx, hipMemcpyAsync(stage, coeffs, n * 8, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_narrow(ctx, (const uint64_t *)stage, (uint32_t *)d_in, n, 0));
    size_t grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    scale_kernel<<<(uint32_t)grid, 256, 0, ctx->stream>>>((const uint32_t *)d_in, (uint32_t *)d_out, n, ctx->fs.F, S);
    HIP_TRY(ctx, hipGetLastError());
    SMI_TRY(launch_widen(ctx, (const uint32_t *)d_out, (uint64_t *)stage, n));
    HIP_TRY(ctx, hipMemcpyAsync(out, stage, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    return check_flag(ctx);
}

// Polynomial::mul (src/univariate/mul.rs:6-29) through two forward transforms, a pointwise
// product and one inverse transform (the reference's schoolbook loop is O(n^2)).
__global__ void pointwise_mul_kernel(uint32_t *a, const uint32_t *b, size_t n, Fp F) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) a[i] = mont_mul(to_mont(a[i], F), b[i], F);
}
int smi_poly_mul(smi_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out, size_t *n_out) {
    if (!ctx || !n_out || (na && !a) || (nb && !b)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    // zero operands give the empty polynomial, exactly like mul.rs:7-12 (is_zero = deg == -1)
    bool za = true, zb = true;
    for (size_t i = 0; i < na; i++) za = za && a[i] == 0;
    for (size_t i = 0; i < nb; i++) zb = zb && b[i] == 0;
    if (za || zb) { *n_out = 0; return SMI_OK; }
    if (!out) return SMI_ERR_BAD_ARG;
    const size_t n = na + nb - 1;   // coeffs.len() of the reference's result
    uint32_t L = 0;
    while (((size_t)1 << L) < n) L++;
    if (L > ctx->fs.K) return smi_fail(ctx, SMI_ERR_ROOT_TOO_LARGE, "product too long for this modulus");
    const size_t N = (size_t)1 << L;
    void *stage, *d_a, *d_b, *d_fa;
    SMI_TRY(ctx_tmp(ctx, 0, (N > na + nb ? N : na + nb) * 8, &stage));
    SMI_TRY(ctx_tmp(ctx, 1, (na + nb) * 4, &d_a));
    SMI_TRY(ctx_tmp(ctx, 2, N * 4, &d_fa));
    SMI_TRY(ctx_tmp(ctx, 3, N * 4, &d_b));
    uint32_t *ua = (uint32_t *)d_a, *ub = ua + na, *fa = (uint32_t *)d_fa, *fb = (uint32_t *)d_b;
    HIP_TRY(ctx, hipMemcpyAsync(stage, a, na * 8, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_narrow(ctx, (const uint64_t *)stage, ua, na, 0));
    HIP_TRY(ctx, hipMemcpyAsync((uint64_t *)stage + na, b, nb * 8, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_narrow(ctx, (const uint64_t *)stage + na, ub, nb, 0));
    SMI_TRY(dev_ntt(ctx, ua, fa, L, na, 1, na, N, 0, 1, 1));
    SMI_TRY(dev_ntt(ctx, ub, fb, L, nb, 1, nb, N, 0, 1, 1));
    size_t grid = (N + 255) / 256;
    if (grid > 2048) grid = 2048;
    pointwise_mul_kernel<<<(uint32_t)grid, 256, 0, ctx->stream>>>(fa, fb, N, ctx->fs.F);
    HIP_TRY(ctx, hipGetLastError());
    SMI_TRY(dev_ntt(ctx, fa, fa, L, N, 1, N, N, 1, 1, 1));
    SMI_TRY(launch_widen(ctx, fa, (uint64_t *)stage, n));
    HIP_TRY(ctx, hipMemcpyAsync(out, stage, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    *n_out = n;
    return check_flag(ctx);
}

// Polynomial::div (src/univariate/div.rs:6-42).  The reference subtracts one shifted multiple of the
// divisor per quotient coefficient (O(n*m) u128 divisions); here the quotient comes from the
// power-series inverse of the reversed divisor (Newton iteration, every product an NTT product on
// the device) and the remainder from one more product:
//     rev(q) = rev(a) * rev(b)^-1  mod x^k,   k = deg a - deg b + 1,      r = a - q*b  (deg r < deg b).
// Quotient and remainder of a division are unique, so the coefficients are the reference's.
__global__ void reverse_kernel(const uint32_t *src, size_t deg, uint32_t *dst, size_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < k; i += step) dst[i] = i <= deg ? src[deg - i] : 0u;   // dst[i] = coefficient deg-i, zero beyond
}
__global__ void two_minus_kernel(uint32_t *e, size_t n, uint32_t p) {   // e <- 2 - e as a power series
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) e[i] = fp_sub(i == 0 ? 2u % p : 0u, e[i], p);
}
__global__ void sub_kernel(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n, uint32_t p) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = fp_sub(a[i], b[i], p);
}
static uint32_t ew_grid(size_t n) { size_t g = (n + 255) / 256; return (uint32_t)(g > 2048 ? 2048 : (g ? g : 1)); }
// first `keep` coefficients of x*y into out (f1, f2: scratch of `cap` elements, cap a power of two >= nx+ny-1)
static int dev_mul_trunc(smi_ctx *ctx, const uint32_t *x, size_t nx, const uint32_t *y, size_t ny, uint32_t *out, size_t keep,
                         uint32_t *f1, uint32_t *f2) {
    const size_t n = nx + ny - 1;
    uint32_t L = 0;
    while (((size_t)1 << L) < n) L++;
    const size_t N = (size_t)1 << L;
    SMI_TRY(dev_ntt(ctx, x, f1, L, nx, 1, nx, N, 0, 1, 1));
    SMI_TRY(dev_ntt(ctx, y, f2, L, ny, 1, ny, N, 0, 1, 1));
    pointwise_mul_kernel<<<ew_grid(N), 256, 0, ctx->stream>>>(f1, f2, N, ctx->fs.F);
    HIP_TRY(ctx, hipGetLastError());
    SMI_TRY(dev_ntt(ctx, f1, f1, L, N, 1, N, N, 1, 1, 1));
    HIP_TRY(ctx, hipMemcpyAsync(out, f1, (keep < n ? keep : n) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (keep > n) HIP_TRY(ctx, hipMemsetAsync(out + n, 0, (keep - n) * 4, ctx->stream));
    return SMI_OK;
}
int smi_poly_div(smi_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *q, size_t *nq, uint64_t *r,
                 size_t *nr) {
    if (!ctx || !nq || !nr || (na && !a) || (nb && !b)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    const uint32_t p = ctx->fs.F.p;
    ptrdiff_t da = -1, db = -1;   // Polynomial::deg (src/univariate/mod.rs:37-46)
    for (size_t i = 0; i < na; i++) { if (a[i] >= p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "coefficient >= p"); if (a[i]) da = (ptrdiff_t)i; }
    for (size_t i = 0; i < nb; i++) { if (b[i] >= p) return smi_fail(ctx, SMI_ERR_NON_CANONICAL, "coefficient >= p"); if (b[i]) db = (ptrdiff_t)i; }
    if (db < 0) return smi_fail(ctx, SMI_ERR_POLY_DIV_BY_ZERO, "No division by zero");
    if (da < db) {   // quotient vec![], remainder numer.clone() (div.rs:10-18)
        *nq = 0;
        *nr = na;
        if (na && !r) return SMI_ERR_BAD_ARG;
        if (na) memcpy(r, a, na * 8);
        return SMI_OK;
    }
    if (!q || (db && !r)) return SMI_ERR_BAD_ARG;
    const size_t k = (size_t)(da - db) + 1, n = (size_t)da + 1, m = (size_t)db + 1;
    size_t cap = 1;
    while (cap < 2 * n) cap <<= 1;   // every product below has fewer than 2n terms
    uint32_t capL = 0;
    while (((size_t)1 << capL) < cap) capL++;
    if (capL > ctx->fs.K) return smi_fail(ctx, SMI_ERR_ROOT_TOO_LARGE, "dividend too long for this modulus");
    uint32_t *buf = nullptr;
    uint64_t *stage = nullptr;
    if (hipMalloc((void **)&buf, (n + 2 * m + 4 * k + 2 * cap) * 4) != hipSuccess || hipMalloc((void **)&stage, n * 8) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(buf);
        return smi_fail(ctx, SMI_ERR_OOM, "poly_div scratch");
    }
    uint32_t *A = buf, *B = A + n, *rA = B + m, *rB = rA + k, *g = rB + k, *e = g + k, *R = e + k, *f1 = R + m, *f2 = f1 + cap;
    int rc = SMI_OK;
    auto run = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(stage, a, n * 8, hipMemcpyHostToDevice, ctx->stream));
        SMI_TRY(launch_narrow(ctx, stage, A, n, 0));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // stage is reused for b
        HIP_TRY(ctx, hipMemcpyAsync(stage, b, m * 8, hipMemcpyHostToDevice, ctx->stream));
        SMI_TRY(launch_narrow(ctx, stage, B, m, 0));
        reverse_kernel<<<ew_grid(k), 256, 0, ctx->stream>>>(A, (size_t)da, rA, k);
        reverse_kernel<<<ew_grid(k), 256, 0, ctx->stream>>>(B, (size_t)db, rB, k);
        // g = rB^-1 mod x^k: g_1 = 1 / lead(b), g_2t = g_t * (2 - rB * g_t) mod x^2t
        const uint32_t g0 = h_inv(ctx, (uint32_t)b[db]);
        HIP_TRY(ctx, hipMemsetAsync(g, 0, k * 4, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(g, &g0, 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // g0 lives on this frame
        for (size_t t = 1; t < k; t <<= 1) {
            const size_t t2 = 2 * t < k ? 2 * t : k;
            SMI_TRY(dev_mul_trunc(ctx, rB, t2, g, t, e, t2, f1, f2));
            two_minus_kernel<<<ew_grid(t2), 256, 0, ctx->stream>>>(e, t2, p);
            SMI_TRY(dev_mul_trunc(ctx, g, t, e, t2, g, t2, f1, f2));
        }
        SMI_TRY(dev_mul_trunc(ctx, rA, k, g, k, e, k, f1, f2));            // rev(q)
        reverse_kernel<<<ew_grid(k), 256, 0, ctx->stream>>>(e, k - 1, rA, k);   // q, in rA
        SMI_TRY(launch_widen(ctx, rA, stage, k));
        HIP_TRY(ctx, hipMemcpyAsync(q, stage, k * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (db) {   // r = a - q*b: the terms of degree >= deg b cancel, db coefficients remain
            SMI_TRY(dev_mul_trunc(ctx, rA, k, B, m, R, (size_t)db, f1, f2));
            sub_kernel<<<ew_grid((size_t)db), 256, 0, ctx->stream>>>(A, R, R, (size_t)db, p);
            SMI_TRY(launch_widen(ctx, R, stage, (size_t)db));
            HIP_TRY(ctx, hipMemcpyAsync(r, stage, (size_t)db * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(ctx, hipGetLastError());
        return check_flag(ctx);
    };
    rc = run();
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(buf);
    (void)hipFree(stage);
    if (rc != SMI_OK) return rc;
    *nq = k;
    *nr = (size_t)db;
    return SMI_OK;
}

int smi_lde(smi_ctx *ctx, const uint64_t *cols, uint32_t n_cols, uint32_t log_n, uint32_t log_blowup, uint64_t trace_offset,
            uint64_t lde_offset, uint64_t *out) {
    if (!ctx || !cols || !out || !n_cols) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (log_n + log_blowup > ctx->fs.K)
        return smi_fail(ctx, ctx->fs.F.p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "LDE domain too large");
    const size_t n = (size_t)1 << log_n, N = n << log_blowup;
    void *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 1, (size_t)n_cols * n * 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, (size_t)n_cols * N * 4, &d_out));
    SMI_TRY(host_to_dev_u32(ctx, cols, (size_t)n_cols * n, (uint32_t *)d_in, 0));
    SMI_TRY(smi_dev_lde(ctx, (const uint32_t *)d_in, n_cols, log_n, log_blowup, trace_offset, lde_offset, (uint32_t *)d_out));
    return dev_u32_to_host(ctx, (const uint32_t *)d_out, (size_t)n_cols * N, out);
}

// Trace::to_field_elements + get_col (src/trace.rs:21-34): row-major i128 -> column-major u64.
// `e as u64` keeps the low 64 bits; the device needs canonical residues, so reduce mod p here.
int smi_trace_pack(const smi_ctx *ctx, const void *rows_i128, size_t n_rows, size_t n_cols, uint64_t *cols_out) {
    if (!ctx || !rows_i128 || !cols_out) return SMI_ERR_BAD_ARG;
    const uint64_t *w = (const uint64_t *)rows_i128;  // little-endian i128 = (lo, hi)
    const uint32_t p = ctx->fs.F.p;
    for (size_t r = 0; r < n_rows; r++)
        for (size_t c = 0; c < n_cols; c++) cols_out[c * n_rows + r] = w[2 * (r * n_cols + c)] % p;
    return SMI_OK;
}

// ------------------------------------------------------------------------- hash / merkle
int smi_dev_hash_leaves(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_digests) {
    if (!ctx || (n && (!d_elems || !d_digests))) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return launch_leaf_hash(ctx, d_elems, n, d_digests);
}
int smi_dev_merkle_build(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes) {
    if (!ctx || !d_elems || !d_nodes) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (n == 0) return smi_fail(ctx, SMI_ERR_EMPTY_LEAVES, nullptr);
    if (!is_pow2(n)) return smi_fail(ctx, SMI_ERR_LEAVES_NOT_POW2, nullptr);
    return launch_merkle(ctx, d_elems, n, d_nodes);
}
int launch_merkle_rows(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t col_stride, size_t n, uint8_t *d_nodes);
int smi_dev_merkle_build_rows(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t col_stride, size_t n, uint8_t *d_nodes) {
    if (!ctx || !d_cols || !d_nodes) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (n == 0) return smi_fail(ctx, SMI_ERR_EMPTY_LEAVES, nullptr);
    if (!is_pow2(n)) return smi_fail(ctx, SMI_ERR_LEAVES_NOT_POW2, nullptr);
    return launch_merkle_rows(ctx, d_cols, n_cols, col_stride, n, d_nodes);
}
int smi_dev_merkle_from_digests(smi_ctx *ctx, size_t n, uint8_t *d_nodes) {
    if (!ctx || !d_nodes) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (n == 0) return smi_fail(ctx, SMI_ERR_EMPTY_LEAVES, nullptr);
    if (!is_pow2(n)) return smi_fail(ctx, SMI_ERR_LEAVES_NOT_POW2, nullptr);
    return launch_merkle(ctx, nullptr, n, d_nodes);
}

int smi_hash_leaves(smi_ctx *ctx, const uint64_t *elems, size_t n, uint8_t *digests) {
    if (!ctx || (n && (!elems || !digests))) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (!n) return SMI_OK;
    void *stage, *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 0, n * 8, &stage));
    SMI_TRY(ctx_tmp(ctx, 1, n * 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, n * 32, &d_out));
    HIP_TRY(ctx, hipMemcpyAsync(stage, elems, n * 8, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_narrow(ctx, (const uint64_t *)stage, (uint32_t *)d_in, n, 0));
    SMI_TRY(launch_leaf_hash(ctx, (const uint32_t *)d_in, n, (uint8_t *)d_out));
    HIP_TRY(ctx, hipMemcpyAsync(digests, d_out, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    return check_flag(ctx);
}
int smi_hash_combine_pairs(smi_ctx *ctx, const uint8_t *digests, size_t n_pairs, uint8_t *out) {
    if (!ctx || (n_pairs && (!digests || !out))) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (!n_pairs) return SMI_OK;
    void *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 1, n_pairs * 64, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, n_pairs * 32, &d_out));
    HIP_TRY(ctx, hipMemcpyAsync(d_in, digests, n_pairs * 64, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_combine(ctx, (const uint8_t *)d_in, n_pairs, (uint8_t *)d_out));
    HIP_TRY(ctx, hipMemcpyAsync(out, d_out, n_pairs * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}
int smi_hash_bytes(smi_ctx *ctx, const uint8_t *msg, size_t len, uint8_t out[32]) {
    if (!ctx || (len && !msg) || !out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    void *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 1, len + 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, 32, &d_out));
    if (len) HIP_TRY(ctx, hipMemcpyAsync(d_in, msg, len, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_hash_bytes(ctx, (const uint8_t *)d_in, len, (uint32_t *)d_out));
    HIP_TRY(ctx, hipMemcpyAsync(out, d_out, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}

int smi_dev_hash_bytes(smi_ctx *ctx, const uint8_t *d_msg, size_t len, uint8_t *d_out32) {
    if (!ctx || (len && !d_msg) || !d_out32) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    return launch_hash_bytes(ctx, d_msg, len, (uint32_t *)d_out32);
}

int launch_hash_bytes_batch(smi_ctx *ctx, const uint8_t *d_msgs, size_t n, size_t len, uint32_t *d_out);
int smi_hash_bytes_batch(smi_ctx *ctx, const uint8_t *msgs, size_t n, size_t msg_len, uint8_t *out) {
    if (!ctx || (n && msg_len && !msgs) || (n && !out)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (!n) return SMI_OK;
    if (n > ((size_t)1 << 24) || msg_len > ((size_t)1 << 20)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "hash_bytes_batch: at most 2^24 messages of 2^20 bytes");
    void *d_in, *d_out;
    SMI_TRY(ctx_tmp(ctx, 1, n * msg_len + 4, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, n * 32, &d_out));
    if (msg_len) HIP_TRY(ctx, hipMemcpyAsync(d_in, msgs, n * msg_len, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_hash_bytes_batch(ctx, (const uint8_t *)d_in, n, msg_len, (uint32_t *)d_out));
    HIP_TRY(ctx, hipMemcpyAsync(out, d_out, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}

static int tree_alloc(smi_ctx *ctx, size_t n, smi_tree **out) {
    if (n == 0) return smi_fail(ctx, SMI_ERR_EMPTY_LEAVES, nullptr);        // src/merkle.rs:12
    if (!is_pow2(n)) return smi_fail(ctx, SMI_ERR_LEAVES_NOT_POW2, nullptr);  // src/merkle.rs:13-16
    smi_tree *t = new smi_tree{ctx, nullptr, n, true};
    if (hipMalloc((void **)&t->d_nodes, (2 * n - 1) * 32) != hipSuccess) {
        delete t;
        return smi_fail(ctx, SMI_ERR_OOM, "hipMalloc tree");
    }
    *out = t;
    return SMI_OK;
}
int smi_merkle_new(smi_ctx *ctx, const uint8_t *leaves, size_t n, smi_tree **out) {
    if (!ctx || !out || (n && !leaves)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    smi_tree *t = nullptr;
    SMI_TRY(tree_alloc(ctx, n, &t));
    int rc = SMI_OK;
    if (hipMemcpyAsync(t->d_nodes, leaves, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = smi_fail(ctx, SMI_ERR_HIP, "upload leaves");
    if (rc == SMI_OK) rc = launch_merkle(ctx, nullptr, n, t->d_nodes);
    if (rc == SMI_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = smi_fail(ctx, SMI_ERR_HIP, "merkle sync");
    if (rc != SMI_OK) {
        smi_merkle_free(t);
        return rc;
    }
    *out = t;
    return SMI_OK;
}
int smi_merkle_from_codeword(smi_ctx *ctx, const uint64_t *codeword, size_t n, smi_tree **out) {
    if (!ctx || !out || (n && !codeword)) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    smi_tree *t = nullptr;
    SMI_TRY(tree_alloc(ctx, n, &t));
    void *d_in;
    int rc = ctx_tmp(ctx, 1, n * 4, &d_in);
    if (rc == SMI_OK) rc = host_to_dev_u32(ctx, codeword, n, (uint32_t *)d_in, 0);
    if (rc == SMI_OK) rc = launch_merkle(ctx, (const uint32_t *)d_in, n, t->d_nodes);
    if (rc == SMI_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = smi_fail(ctx, SMI_ERR_HIP, "merkle build");
    if (rc != SMI_OK) {
        smi_merkle_free(t);
        return rc;
    }
    *out = t;
    return SMI_OK;
}
int smi_merkle_commit(smi_ctx *ctx, const uint8_t *leaves, size_t n, uint8_t root[32]) {
    if (!ctx || !root) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    smi_tree *t = nullptr;
    SMI_TRY(smi_merkle_new(ctx, leaves, n, &t));
    int rc = smi_merkle_root(ctx, t, root);
    smi_merkle_free(t);
    return rc;
}
int smi_merkle_root(smi_ctx *ctx, const smi_tree *t, uint8_t root[32]) {
    if (!ctx || !t || !root) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(root, t->d_nodes + (2 * t->n - 2) * 32, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}
int smi_merkle_open(smi_ctx *ctx, const smi_tree *t, size_t index, uint8_t *path, size_t *depth) {
    if (!ctx || !t || !path || !depth) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (index >= t->n) return smi_fail(ctx, SMI_ERR_INDEX_OOB, nullptr);  // src/merkle.rs:68
    size_t idx = index, off = 0, len = t->n, d = 0;
    while (len > 1) {  // sibling at each level (src/merkle.rs:73-77)
        HIP_TRY(ctx, hipMemcpyAsync(path + 32 * d, t->d_nodes + (off + (idx ^ 1)) * 32, 32, hipMemcpyDeviceToHost, ctx->stream));
        idx >>= 1;
        off += len;
        len >>= 1;
        d++;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *depth = d;
    return SMI_OK;
}
int smi_merkle_level(smi_ctx *ctx, const smi_tree *t, uint32_t level, uint8_t *out, size_t *n_out) {
    if (!ctx || !t || !out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (level > ilog2(t->n)) return smi_fail(ctx, SMI_ERR_INDEX_OOB, nullptr);
    const size_t cnt = t->n >> level, off = 2 * t->n - ((2 * t->n) >> level);
    HIP_TRY(ctx, hipMemcpyAsync(out, t->d_nodes + off * 32, cnt * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_out) *n_out = cnt;
    return SMI_OK;
}
int smi_merkle_verify_batch(smi_ctx *ctx, const uint8_t *leaves, const uint64_t *indices, const uint8_t *paths, size_t k, size_t depth,
                            const uint8_t root[32], uint8_t *ok) {
    if (!ctx || !root || (k && (!leaves || !indices || !ok || (depth && !paths)))) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(ctx);
    if (!k) return SMI_OK;
    if (depth > 64) return smi_fail(ctx, SMI_ERR_BAD_ARG, "verify: path deeper than 64");
    const size_t b_leaves = k * 32, b_idx = k * 8, b_paths = k * depth * 32;
    void *d_in, *d_ok;
    SMI_TRY(ctx_tmp(ctx, 1, b_leaves + b_idx + b_paths + 32 + 64, &d_in));
    SMI_TRY(ctx_tmp(ctx, 2, k, &d_ok));
    uint8_t *base = (uint8_t *)d_in, *dl = base, *di = dl + b_leaves, *dp = di + b_idx, *dr = dp + ((b_paths + 15) & ~(size_t)15);
    HIP_TRY(ctx, hipMemcpyAsync(dl, leaves, b_leaves, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(di, indices, b_idx, hipMemcpyHostToDevice, ctx->stream));
    if (b_paths) HIP_TRY(ctx, hipMemcpyAsync(dp, paths, b_paths, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dr, root, 32, hipMemcpyHostToDevice, ctx->stream));
    SMI_TRY(launch_verify_paths(ctx, dl, (const uint64_t *)di, dp, k, (uint32_t)depth, dr, (uint8_t *)d_ok));
    HIP_TRY(ctx, hipMemcpyAsync(ok, d_ok, k, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SMI_OK;
}
size_t smi_merkle_num_leaves(const smi_tree *t) { return t ? t->n : 0; }
void smi_merkle_free(smi_tree *t) {
    DeviceGuard dg__(t ? t->ctx : nullptr);
    if (!t) return;
    if (t->ctx && t->ctx->stream) (void)hipStreamSynchronize(t->ctx->stream);
    if (t->owns) (void)hipFree(t->d_nodes);
    delete t;
}
void smi_free(void *p) { free(p); }
