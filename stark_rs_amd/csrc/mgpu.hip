// mgpu.hip -- multi-GPU entry points of libstarkmi.so: Fri::commit / Fri::prove
// (reference src/fri.rs:105-156, 250-311) over one codeword sharded across the GPUs of a
// node, and the build-defined trace -> proof composition of stark.hip over them, behind the C
// ABI (include/stark_mi.h, smi_mgpu_*).  One process per GPU.
//
// The round loop is mgpu_loop.h (device-agnostic C++); this file supplies
//   * HipDev   -- its device work: the single-GPU engine's kernels on the context's stream, plus
//                 the sharded query kernel and the coset interleave below;
//   * RcclColl -- its collectives as RCCL calls enqueued on the same stream (all-gather of
//                 sub-roots, grouped send/recv for the fold exchange and the extension's
//                 all-to-all, one byte-sum all-reduce of the proof): nothing in a prove waits for
//                 the host until the proof is copied back;
//   * CallbackColl -- the same three operations through caller-supplied functions (tests on one
//                 GPU, other transports).
// librccl is opened with dlopen when the first communicator is created, so single-GPU users of
// the library never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "internal.h"
#include "mgpu_loop.h"

// launchers of the single-GPU engine (hash.hip, fri.hip, stark.hip)
int launch_merkle(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes);
int launch_merkle_fs(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t *fs_words, uint8_t *proof_slot,
                     uint64_t *alpha_out, bool *done);
int launch_merkle_batch(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes, uint32_t n_trees, size_t elem_stride,
                        size_t node_stride_bytes, uint32_t row_cols = 0, size_t row_stride = 0);
int launch_fold_shard(smi_ctx *ctx, const uint32_t *d_lo, const uint32_t *d_hi, size_t count, size_t i0, size_t full_len,
                      const uint64_t *d_alpha, uint64_t offset, uint64_t omega, uint32_t *d_out);
size_t fri_fs_bytes();
int launch_fs_init(smi_ctx *ctx, void *fs);
int launch_fs_round(smi_ctx *ctx, void *fs, const uint8_t *root, uint8_t *proof_slot, uint64_t *alpha_out);
int launch_fs_challenge(smi_ctx *ctx, const void *fs, uint64_t *out);
int launch_sample_indices(smi_ctx *ctx, const uint64_t *challenge, uint64_t size, uint64_t reduced_size, uint32_t number,
                          uint64_t *indices, uint64_t *reduced);
int launch_emit_codeword(smi_ctx *ctx, const uint32_t *cw, uint64_t len, uint8_t *dst);
int launch_fs_weights(smi_ctx *ctx, const uint8_t *const *d_root_ptrs, uint32_t n, uint64_t *weights, uint8_t *roots_out);
int launch_column_open(smi_ctx *ctx, const MgSide *d_cols, uint32_t W, const uint64_t *d_top, uint32_t t, int rank, uint8_t *d_out);

// ------------------------------------------------------------------------- kernels
// Fri::query for every (test, layer): each rank writes what it owns (mgpu_core.h)
__global__ __launch_bounds__(64) void mg_query_kernel(const MgLayer *layers, const uint64_t *top, int rank, uint8_t *proof) {
    const MgLayer L = layers[blockIdx.y];
    mg_query_write(L, top[blockIdx.x], blockIdx.x, rank, proof, threadIdx.x, 64);
}

// out[c][(q << log_b) + r] = in[(c << log_b) + r][q]: the 2^log_b coset planes of a column, as
// they arrive from the ranks that computed them, become its natural-order block.  LDS-tiled:
// coalesced along q on the way in, along the output on the way out.
__global__ __launch_bounds__(256) void mg_interleave_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t log_b,
                                                            size_t nq) {
    __shared__ uint32_t t[16][257];
    const uint32_t B = 1u << log_b, tid = threadIdx.x;
    const size_t q0 = (size_t)blockIdx.x * 256, c = blockIdx.y;
    const uint32_t *in_c = in + (c << log_b) * nq;
    uint32_t *out_c = out + ((c * nq) << log_b);
    for (uint32_t r = 0; r < B; r++)
        if (q0 + tid < nq) t[r][tid] = in_c[r * nq + q0 + tid];
    __syncthreads();
    for (uint32_t i = 0; i < B; i++) {
        const uint32_t j = i * 256 + tid, q = j >> log_b, r = j & (B - 1);
        if (q0 + q < nq) out_c[((q0 + q) << log_b) + r] = t[r][q];
    }
}

// ------------------------------------------------------------------------- device
namespace {
struct HipDev : MgDev {
    smi_ctx *ctx;
    explicit HipDev(smi_ctx *c) : ctx(c) {}
    uint32_t prime() const override { return ctx->fs.F.p; }
    uint32_t root_of_unity(uint32_t log_n) const override { return h_root(ctx, log_n); }
    int fail(int code, const char *msg) override { return smi_fail(ctx, code, msg); }
    int reset() override { return arena_reset(ctx); }
    void *alloc(size_t bytes) override { return arena_alloc(ctx, bytes ? bytes : 4); }
    int copy(void *dst, const void *src, size_t bytes) override {
        if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return SMI_OK;
    }
    int copy_rows(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, size_t rows) override {
        if (row_bytes && rows)
            HIP_TRY(ctx, hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, row_bytes, rows, hipMemcpyDeviceToDevice, ctx->stream));
        return SMI_OK;
    }
    int zero(void *dst, size_t bytes) override {
        if (bytes) HIP_TRY(ctx, hipMemsetAsync(dst, 0, bytes, ctx->stream));
        return SMI_OK;
    }
    int upload(void *dst, const void *host, size_t bytes) override {
        if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SMI_OK;
    }
    int download(void *host, const void *src, size_t bytes) override {
        if (bytes) HIP_TRY(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return SMI_OK;
    }
    int sync() override {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return SMI_OK;
    }
    int merkle(const uint32_t *elems, size_t n, uint8_t *nodes) override { return launch_merkle(ctx, elems, n, nodes); }
    int merkle_fs(const uint32_t *elems, size_t n, uint8_t *nodes, void *fs, uint8_t *proof_slot, uint64_t *alpha_out) override {
        bool done = false;   // the launch that finishes the tree runs the Fiat-Shamir round when it can (hash.hip)
        SMI_TRY(launch_merkle_fs(ctx, elems, n, nodes, (uint32_t *)fs, proof_slot, alpha_out, &done));
        if (!done) SMI_TRY(launch_fs_round(ctx, fs, nodes + (2 * n - 2) * 32, proof_slot, alpha_out));
        return SMI_OK;
    }
    int merkle_batch(const uint32_t *elems, size_t n, uint8_t *nodes, uint32_t n_trees, size_t elem_stride, size_t node_stride_bytes) override {
        return launch_merkle_batch(ctx, elems, n, nodes, n_trees, elem_stride, node_stride_bytes);
    }
    int merkle_from_digests(size_t n, uint8_t *nodes) override { return launch_merkle(ctx, nullptr, n, nodes); }
    size_t fs_bytes() const override { return fri_fs_bytes(); }
    int fs_init(void *fs) override { return launch_fs_init(ctx, fs); }
    int fs_round(void *fs, const uint8_t *root, uint8_t *proof_slot, uint64_t *alpha_out) override {
        return launch_fs_round(ctx, fs, root, proof_slot, alpha_out);
    }
    int fs_challenge(const void *fs, uint64_t *out) override { return launch_fs_challenge(ctx, fs, out); }
    int fs_weights(const uint8_t *const *d_root_ptrs, uint32_t n, uint64_t *weights, uint8_t *roots_out) override {
        return launch_fs_weights(ctx, d_root_ptrs, n, weights, roots_out);
    }
    int fold_shard(const uint32_t *lo, const uint32_t *hi, size_t count, size_t i0, size_t full_len, const uint64_t *alpha, uint64_t offset,
                   uint64_t omega, uint32_t *out) override {
        return launch_fold_shard(ctx, lo, hi, count, i0, full_len, alpha, offset, omega, out);
    }
    // the replicated rounds' tail in ONE workgroup launch, as in the single-GPU fri_run (csrc/hash.hip, fri_tail_kernel)
    uint64_t tail_max_len() const override { return fri_tail_len(); }
    uint32_t tail_max_rounds() const override { return SMI_FRI_TAIL_MAX_ROUNDS; }
    int fri_tail(const MgTailRound *rounds, uint32_t n_rounds, void *fs) override {
        if (!n_rounds) return SMI_OK;
        ScaleScope pin__(ctx);   // one x^-1 table per fold, all held until the single launch
        FriTailArgs ta;
        memset(&ta, 0, sizeof ta);
        ta.n_rounds = n_rounds;
        ta.fs_words = (uint32_t *)fs;
        ta.F = ctx->fs.F;
        ta.inv2_m = (uint32_t)(((uint64_t)h_inv(ctx, 2) << 32) % ctx->fs.F.p);
        for (uint32_t k = 0; k < n_rounds; k++) {
            const MgTailRound &t = rounds[k];
            FriTailRound &tr = ta.r[k];
            tr.cw = t.cw; tr.next = t.next; tr.nodes = t.nodes; tr.proof_slot = t.proof_slot; tr.alpha_out = t.alpha_out;
            tr.len = (uint32_t)t.len;
            if (!t.next) break;
            if (t.len < 2) return smi_fail(ctx, SMI_ERR_BAD_ARG, "fold: codeword length must be a power of two >= 2");
            if (t.offset == 0 || t.omega == 0) return smi_fail(ctx, SMI_ERR_DIV_BY_ZERO, "no division by zero");   // src/ff.rs:182
            uint32_t lg = 0;
            while ((2ull << lg) < t.len) lg++;
            SMI_TRY(ctx_scale_tables(ctx, h_inv(ctx, (uint32_t)t.offset), h_inv(ctx, (uint32_t)t.omega), lg, &tr.S));
        }
        return launch_fri_tail(ctx, ta);
    }
    int emit_codeword(const uint32_t *cw, uint64_t len, uint8_t *dst) override { return launch_emit_codeword(ctx, cw, len, dst); }
    int sample_indices(const uint64_t *challenge, uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t *indices,
                       uint64_t *reduced) override {
        return launch_sample_indices(ctx, challenge, size, reduced_size, number, indices, reduced);
    }
    int query(const MgLayer *layers_host, uint32_t n_layers, const uint64_t *top, uint32_t t, int rank, uint8_t *proof) override {
        MgLayer *d_layers = (MgLayer *)alloc(sizeof(MgLayer) * n_layers);
        if (!d_layers) return fail(SMI_ERR_OOM, "mgpu: layer table");
        HIP_TRY(ctx, hipMemcpyAsync(d_layers, layers_host, sizeof(MgLayer) * n_layers, hipMemcpyHostToDevice, ctx->stream));
        mg_query_kernel<<<dim3(t, n_layers), 64, 0, ctx->stream>>>(d_layers, top, rank, proof);
        HIP_TRY(ctx, hipGetLastError());
        return SMI_OK;
    }
    int column_open(const MgSide *cols_host, uint32_t W, const uint64_t *top, uint32_t t, int rank, uint8_t *out) override {
        MgSide *d_cols = (MgSide *)alloc(sizeof(MgSide) * W);
        if (!d_cols) return fail(SMI_ERR_OOM, "mgpu: column table");
        HIP_TRY(ctx, hipMemcpyAsync(d_cols, cols_host, sizeof(MgSide) * W, hipMemcpyHostToDevice, ctx->stream));
        return launch_column_open(ctx, d_cols, W, top, t, rank, out);
    }
    int lde(const uint32_t *trace, uint32_t n_cols, uint32_t log_n, uint32_t log_b, uint64_t trace_offset, uint64_t lde_offset,
            uint32_t *out) override {
        return smi_dev_lde(ctx, trace, n_cols, log_n, log_b, trace_offset, lde_offset, out);
    }
    int ntt(const uint32_t *in, uint32_t *out, uint32_t log_n, size_t n_in, uint32_t batch, size_t in_stride, size_t out_stride, int inverse,
            uint64_t offset, uint64_t post_scale) override {
        return dev_ntt(ctx, in, out, log_n, n_in, batch, in_stride, out_stride, inverse, offset, post_scale);
    }
    int ntt_shard_first(uint32_t *strip, uint32_t log_n, uint32_t log_g, uint32_t rank, int inverse, uint64_t offset) override {
        return dev_ntt_shard_first(ctx, strip, log_n, log_g, rank, inverse, offset);
    }
    int ntt_shard_rest(uint32_t *rows, uint32_t *out, uint32_t log_n, uint32_t log_g, int inverse) override {
        return dev_ntt_shard_rest(ctx, rows, out, log_n, log_g, inverse);
    }
    int interleave(const uint32_t *in, uint32_t *out, uint32_t n_cols, uint32_t log_b, size_t nq) override {
        if (log_b > 4) return fail(SMI_ERR_BAD_ARG, "mgpu: blowup above 16 is not sharded");
        if (!n_cols || !nq) return SMI_OK;
        mg_interleave_kernel<<<dim3((uint32_t)((nq + 255) / 256), n_cols), 256, 0, ctx->stream>>>(in, out, log_b, nq);
        HIP_TRY(ctx, hipGetLastError());
        return SMI_OK;
    }
    int combine(const uint32_t *cols, uint32_t n_cols, size_t len, size_t stride, const uint64_t *weights, uint32_t *out) override {
        return smi_dev_combine_columns(ctx, cols, n_cols, len, stride, weights, out);
    }
};

// ------------------------------------------------------------------------- RCCL
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi *rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *override_path = getenv("SMI_RCCL_LIB");
        const char *names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm) continue;
            if ((api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if (api.handle) {
#define SYM(f) api.f = (decltype(api.f))dlsym(api.handle, "nccl" #f)
            SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(AllGather); SYM(AllReduce);
            SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.Send || !api.Recv ||
                !api.GroupStart || !api.GroupEnd) {
                dlclose(api.handle);
                api.handle = nullptr;
            }
        }
    }
    return api.handle ? &api : nullptr;
}

struct RcclColl : MgColl {
    smi_ctx *ctx;
    RcclApi *api;
    ncclComm_t comm = nullptr;
    RcclColl(smi_ctx *c, RcclApi *a) : ctx(c), api(a) {}
    ~RcclColl() override {
        if (comm) (void)api->CommDestroy(comm);
    }
    int check(ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return SMI_OK;
        ctx->err = std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
        return SMI_ERR_RCCL;
    }
    bool stream_ordered() const override { return true; }
    int all_gather(const void *send, void *recv, size_t bytes_per_rank) override {
        return check(api->AllGather(send, recv, bytes_per_rank, ncclUint8, comm, ctx->stream), "ncclAllGather");
    }
    int exchange(const std::vector<MgXfer> &sends, const std::vector<MgXfer> &recvs) override {
        if (sends.empty() && recvs.empty()) return SMI_OK;
        SMI_TRY(check(api->GroupStart(), "ncclGroupStart"));
        int rc = SMI_OK;
        for (const MgXfer &x : sends)
            if (rc == SMI_OK) rc = check(api->Send(x.ptr, x.bytes, ncclUint8, x.peer, comm, ctx->stream), "ncclSend");
        for (const MgXfer &x : recvs)
            if (rc == SMI_OK) rc = check(api->Recv(x.ptr, x.bytes, ncclUint8, x.peer, comm, ctx->stream), "ncclRecv");
        const int rc2 = check(api->GroupEnd(), "ncclGroupEnd");
        return rc != SMI_OK ? rc : rc2;
    }
    int all_reduce_sum_u8(void *buf, size_t bytes) override {
        return check(api->AllReduce(buf, buf, bytes, ncclUint8, ncclSum, comm, ctx->stream), "ncclAllReduce");
    }
};

struct CallbackColl : MgColl {
    smi_ctx *ctx;
    smi_mgpu_coll ops;
    CallbackColl(smi_ctx *c, const smi_mgpu_coll &o) : ctx(c), ops(o) {}
    bool stream_ordered() const override { return false; }
    int done(int rc, const char *what) { return rc == 0 ? SMI_OK : smi_fail(ctx, SMI_ERR_RCCL, what); }
    int all_gather(const void *send, void *recv, size_t bytes_per_rank) override {
        return done(ops.all_gather(ops.user, send, recv, bytes_per_rank), "collective shim: all_gather failed");
    }
    int exchange(const std::vector<MgXfer> &sends, const std::vector<MgXfer> &recvs) override {
        std::vector<int> sp, rp;
        std::vector<void *> sptr, rptr;
        std::vector<size_t> sb, rb;
        for (const MgXfer &x : sends) { sp.push_back(x.peer); sptr.push_back(x.ptr); sb.push_back(x.bytes); }
        for (const MgXfer &x : recvs) { rp.push_back(x.peer); rptr.push_back(x.ptr); rb.push_back(x.bytes); }
        return done(ops.exchange(ops.user, (int)sends.size(), sp.data(), sptr.data(), sb.data(), (int)recvs.size(), rp.data(), rptr.data(),
                                 rb.data()),
                    "collective shim: exchange failed");
    }
    int all_reduce_sum_u8(void *buf, size_t bytes) override {
        return done(ops.all_reduce_sum_u8(ops.user, buf, bytes), "collective shim: all_reduce failed");
    }
};
}  // namespace

struct smi_mgpu {
    smi_ctx *ctx;
    int rank, world;
    MgColl *coll;
    HipDev dev;
    size_t min_block;
    smi_mgpu(smi_ctx *c, int r, int w, MgColl *k) : ctx(c), rank(r), world(w), coll(k), dev(c), min_block((size_t)1 << 18) {}
};

// ------------------------------------------------------------------------- C ABI
static bool world_ok(int rank, int world) { return world >= 1 && !(world & (world - 1)) && rank >= 0 && rank < world; }

int smi_mgpu_unique_id(uint8_t id[SMI_MGPU_ID_BYTES]) {
    if (!id) return SMI_ERR_BAD_ARG;
    RcclApi *api = rccl_api();
    if (!api) return SMI_ERR_RCCL;
    static_assert(sizeof(ncclUniqueId) <= SMI_MGPU_ID_BYTES, "unique id size");
    ncclUniqueId u;
    memset(id, 0, SMI_MGPU_ID_BYTES);
    if (api->GetUniqueId(&u) != ncclSuccess) return SMI_ERR_RCCL;
    memcpy(id, &u, sizeof u);
    return SMI_OK;
}

int smi_mgpu_create(smi_ctx *ctx, const uint8_t id[SMI_MGPU_ID_BYTES], int rank, int world, smi_mgpu **out) {
    if (!ctx || !id || !out) return SMI_ERR_BAD_ARG;
    *out = nullptr;
    if (!world_ok(rank, world)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "mgpu: world size must be a power of two, 0 <= rank < world");
    RcclApi *api = rccl_api();
    if (!api) return smi_fail(ctx, SMI_ERR_RCCL, "librccl.so could not be loaded (set SMI_RCCL_LIB)");
    DeviceGuard dg__(ctx);
    RcclColl *coll = new RcclColl(ctx, api);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const int rc = coll->check(api->CommInitRank(&coll->comm, world, u, rank), "ncclCommInitRank");
    if (rc != SMI_OK) {
        coll->comm = nullptr;
        delete coll;
        return rc;
    }
    *out = new smi_mgpu(ctx, rank, world, coll);
    return SMI_OK;
}

int smi_mgpu_create_with(smi_ctx *ctx, const smi_mgpu_coll *ops, int rank, int world, smi_mgpu **out) {
    if (!ctx || !ops || !out || !ops->all_gather || !ops->exchange || !ops->all_reduce_sum_u8) return SMI_ERR_BAD_ARG;
    *out = nullptr;
    if (!world_ok(rank, world)) return smi_fail(ctx, SMI_ERR_BAD_ARG, "mgpu: world size must be a power of two, 0 <= rank < world");
    *out = new smi_mgpu(ctx, rank, world, new CallbackColl(ctx, *ops));
    return SMI_OK;
}

void smi_mgpu_destroy(smi_mgpu *m) {
    if (!m) return;
    DeviceGuard dg__(m->ctx);
    (void)hipStreamSynchronize(m->ctx->stream);
    delete m->coll;
    delete m;
}

int smi_mgpu_set_min_block(smi_mgpu *m, size_t min_block) {
    if (!m) return SMI_ERR_BAD_ARG;
    m->min_block = min_block < 2 ? 2 : min_block;
    return SMI_OK;
}

static int give_proof(smi_ctx *ctx, const std::vector<uint8_t> &bytes, uint8_t **proof, size_t *proof_len) {
    *proof = (uint8_t *)malloc(bytes.size() ? bytes.size() : 1);
    if (!*proof) return smi_fail(ctx, SMI_ERR_OOM, "malloc proof");
    memcpy(*proof, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    return SMI_OK;
}

int smi_mgpu_fri_commit(smi_mgpu *m, const smi_fri_cfg *cfg, const uint32_t *d_block, size_t block_len, uint8_t *roots, uint64_t *alphas,
                        uint64_t *last_codeword, size_t *last_len) {
    if (!m || !cfg || !d_block) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(m->ctx);
    SMI_TRY(m->dev.reset());
    MgFriOut o;
    SMI_TRY(mg_fri_run(m->dev, *m->coll, m->rank, m->world, *cfg, d_block, block_len, m->min_block, false, o));
    if (roots)
        for (uint64_t r = 0; r < o.rounds; r++) memcpy(roots + 32 * r, o.proof.data() + 33 * r + 1, 32);
    if (alphas && !o.alphas.empty()) memcpy(alphas, o.alphas.data(), 8 * o.alphas.size());
    if (last_codeword) memcpy(last_codeword, o.proof.data() + 33 * o.rounds + 9, 8 * o.last_len);
    if (last_len) *last_len = o.last_len;
    return SMI_OK;
}

int smi_mgpu_fri_prove(smi_mgpu *m, const smi_fri_cfg *cfg, const uint32_t *d_block, size_t block_len, uint8_t **proof, size_t *proof_len,
                       uint64_t *top_indices) {
    if (!m || !cfg || !d_block || !proof || !proof_len) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(m->ctx);
    SMI_TRY(m->dev.reset());
    MgFriOut o;
    SMI_TRY(mg_fri_run(m->dev, *m->coll, m->rank, m->world, *cfg, d_block, block_len, m->min_block, true, o));
    if (top_indices && !o.top.empty()) memcpy(top_indices, o.top.data(), 8 * o.top.size());
    return give_proof(m->ctx, o.proof, proof, proof_len);
}

int smi_mgpu_lde(smi_mgpu *m, const uint32_t *d_trace_cols, uint32_t n_cols, uint32_t log_n, uint32_t log_blowup, uint64_t trace_offset,
                 uint64_t lde_offset, uint32_t *d_out_blocks) {
    if (!m || !d_trace_cols || !d_out_blocks || !n_cols) return SMI_ERR_BAD_ARG;
    if (log_n + log_blowup > m->ctx->fs.K)
        return smi_fail(m->ctx, m->ctx->fs.F.p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "LDE domain too large");
    DeviceGuard dg__(m->ctx);
    SMI_TRY(m->dev.reset());
    uint32_t *blocks = nullptr;
    size_t stride = 0;
    // the entry point of its own always shards (shard_from = 2): its caller asked for the sharded extension
    SMI_TRY(mg_lde_blocks(m->dev, *m->coll, m->rank, m->world, d_trace_cols, n_cols, log_n, log_blowup, trace_offset, lde_offset, &blocks,
                          &stride, 2));
    const size_t blk = ((size_t)1 << (log_n + log_blowup)) / (size_t)m->world;
    return m->dev.copy_rows(d_out_blocks, blk * 4, blocks, stride * 4, blk * 4, n_cols);
}

int smi_mgpu_ntt(smi_mgpu *m, uint32_t *d_strip, uint32_t *d_out, uint32_t log_n, int inverse, uint64_t offset) {
    if (!m || !d_strip || !d_out || d_strip == d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(m->ctx);
    SMI_TRY(m->dev.reset());
    return mg_ntt(m->dev, *m->coll, m->rank, m->world, d_strip, d_out, log_n, inverse, offset);
}
int smi_mgpu_ntt_natural(smi_mgpu *m, uint32_t *d_strip, uint32_t *d_out, uint32_t log_n, int inverse, uint64_t offset) {
    if (!m || !d_strip || !d_out || d_strip == d_out) return SMI_ERR_BAD_ARG;
    DeviceGuard dg__(m->ctx);
    SMI_TRY(m->dev.reset());
    return mg_ntt(m->dev, *m->coll, m->rank, m->world, d_strip, d_out, log_n, inverse, offset, true);
}
int smi_mgpu_ntt_first_digit(uint32_t log_n, uint32_t *log_r0) {
    if (!log_r0) return SMI_ERR_BAD_ARG;
    const NttPlan pl = ntt_make_plan(log_n, 1);
    if (pl.np < 2) return SMI_ERR_BAD_ARG;
    *log_r0 = (uint32_t)pl.logr[0];
    return SMI_OK;
}

int smi_mgpu_stark_prove(smi_mgpu *m, const smi_stark_cfg *cfg, const uint32_t *d_trace_cols, uint8_t *column_roots, uint8_t **proof,
                         size_t *proof_len, uint64_t *top_indices) {
    if (!m || !cfg || !d_trace_cols || !proof || !proof_len) return SMI_ERR_BAD_ARG;
    if (cfg->log_n + cfg->log_blowup > m->ctx->fs.K)
        return smi_fail(m->ctx, m->ctx->fs.F.p == 998244353u ? SMI_ERR_ROOT_TOO_LARGE : SMI_ERR_UNSUPPORTED_PRIME, "LDE domain too large");
    DeviceGuard dg__(m->ctx);
    MgStarkOut o;
    SMI_TRY(mg_stark_prove(m->dev, *m->coll, m->rank, m->world, *cfg, d_trace_cols, m->min_block, o));
    if (column_roots) memcpy(column_roots, o.column_roots.data(), o.column_roots.size());
    if (top_indices && !o.fri.top.empty()) memcpy(top_indices, o.fri.top.data(), 8 * o.fri.top.size());
    return give_proof(m->ctx, o.fri.proof, proof, proof_len);
}
