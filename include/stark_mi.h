/*
 * stark_mi.h -- C ABI of libstarkmi.so, the MI355X (gfx950) engine that stands behind
 * stark-rs's univariate / trace / fri / merkle API.
 *
 * The reference (0xSooki/stark-rs @ 2026-01-02) has no FFI of its own:
 * each entry point below names the Rust inherent method it replaces (file:line under
 * /root/reference) -- that method's body becomes a call to this function in the
 * binding shown in INTEGRATION.md.
 *
 * Conventions
 *   - Field values cross the boundary as contiguous little-endian u64, the reference's
 *     wire width (src/stream.rs:45, src/hash.rs:33).  They must be canonical (< p)
 *     unless a parameter says "unreduced ok" (SURVEY H6).
 *   - Digests are 32 raw bytes (src/hash.rs:1-2).
 *   - Every function returns an int32 status: 0 = ok, negative = the reference's panic
 *     (one code per message, smi_status_string gives the identical text so the Rust
 *     wrapper can `panic!` with it), <= -100 = HIP/runtime failure (smi_last_error).
 *   - No C++ exceptions, torch types or caller pointers retained across calls.
 *   - Functions named smi_dev_* take device pointers (u32 canonical residues, 4 B per
 *     element on device) and enqueue on the context's stream without synchronising.
 *     All others take host buffers and are synchronous on return.
 *   - A context is single-owner (not thread-safe), one per GPU / process.
 */
#ifndef STARK_MI_H
#define STARK_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: the reference's panic messages ------------------------------- */
enum {
    SMI_OK = 0,
    SMI_ERR_NO_INVERSE = -1,          /* "no inverse"                         src/ff.rs:171 */
    SMI_ERR_DIV_BY_ZERO = -2,         /* "no division by zero"                src/ff.rs:182 */
    SMI_ERR_NOT_POW2 = -3,            /* "n must be a power of two"           src/ff.rs:217 */
    SMI_ERR_ROOT_TOO_LARGE = -4,      /* "n > 2^23 not supported by this modulus" src/ff.rs:218 */
    SMI_ERR_EMPTY_LEAVES = -5,        /* "Cannot create tree from empty leaves"   src/merkle.rs:12 */
    SMI_ERR_LEAVES_NOT_POW2 = -6,     /* "Number of leaves must be power of 2"    src/merkle.rs:13-16 */
    SMI_ERR_INDEX_OOB = -7,           /* "Index out of bounds"                src/merkle.rs:68 */
    SMI_ERR_DOMAIN_NOT_POW2 = -8,     /* "Domain length must be power of 2"   src/fri.rs:37-40 */
    SMI_ERR_EXPANSION_NOT_POW2 = -9,  /* "Expansion factor must be power of 2" src/fri.rs:41-44 */
    SMI_ERR_EXPANSION_TOO_SMALL = -10,/* "Expansion factor must be at least 4" src/fri.rs:45 */
    SMI_ERR_CODEWORD_LEN = -11,       /* "initial codeword length does not match domain length" src/fri.rs:256-260 */
    SMI_ERR_SAMPLE_ENTROPY = -12,     /* "not enough entropy in indices wrt last codeword" src/fri.rs:183-186 */
    SMI_ERR_SAMPLE_TOO_MANY = -13,    /* "cannot sample more indices than available in last codeword" src/fri.rs:187-192 */
    SMI_ERR_LEN_MISMATCH = -14,       /* assert!(domain.len() == values.len()) src/univariate/interpolate.rs:10 */
    SMI_ERR_EMPTY_DOMAIN = -15,       /* assert!(domain.len() > 0)            src/univariate/interpolate.rs:11 */
    SMI_ERR_WRONG_FIELD = -16,        /* assert!(self.p == 998244353)         src/ff.rs:192,216 */
    SMI_ERR_POLY_DIV_BY_ZERO = -18,   /* "No division by zero"                src/univariate/div.rs:7-9 */
    SMI_ERR_NO_ROUNDS = -17,          /* num_rounds()==0: proof the reference's verify rejects */
    /* contract violations that have no reference counterpart */
    SMI_ERR_BAD_ARG = -50,
    SMI_ERR_NON_CANONICAL = -51,      /* a field value >= p where the precondition forbids it (H6) */
    SMI_ERR_UNSUPPORTED_PRIME = -52,  /* p must be an odd prime < 2^30 with p-1 divisible by the sizes used */
    SMI_ERR_NOT_GEOMETRIC = -53,      /* domain is not offset*omega^k: caller must fall back to the CPU code */
    SMI_ERR_COLUMNS_NOT_BOUND = -54,  /* smi_stark_verify on a proof made without open_columns: nothing in it refers to the
                                         column roots, so it can only be checked as a FRI proof (smi_fri_verify) */
    /* runtime */
    SMI_ERR_HIP = -100,
    SMI_ERR_NO_DEVICE = -101,
    SMI_ERR_OOM = -102,
    SMI_ERR_RCCL = -103               /* RCCL (or the caller's collective shim) failed: smi_last_error */
};

typedef struct smi_ctx smi_ctx;
typedef struct smi_tree smi_tree;     /* device-resident MerkleTree (all levels) */
typedef struct smi_fri_run smi_fri_run; /* device-resident artefacts of one Fri::commit */

const char *smi_status_string(int status);
const char *smi_last_error(const smi_ctx *ctx);
const char *smi_version(void);

/* ---- context -------------------------------------------------------------------- */
/* FiniteField::new(p) (src/ff.rs:109-111) plus the generator g() (src/ff.rs:191-197):
 * (998244353, 3) is the reference field; (469762049, 3) is the build's second prime
 * for domains above 2^23.  device = HIP device ordinal. */
int smi_ctx_create(uint64_t p, uint64_t g, int device, smi_ctx **out);
void smi_ctx_destroy(smi_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) for everything enqueued. */
int smi_ctx_set_stream(smi_ctx *ctx, void *hip_stream);
int smi_ctx_sync(smi_ctx *ctx);
/* Per-kernel timing with HIP events on the context's stream (bench.py's roofline leg):
 * while enabled, every hot-path kernel launch is bracketed by two events.
 * smi_ctx_profile_read synchronises, aggregates by kernel name and clears the log. */
typedef struct {
    char name[56];
    uint32_t launches;
    double total_ms;
    double alg_bytes;   /* algorithmic bytes summed over the launches (DESIGN.md states the per-unit figures) */
    double alg_mixes;   /* hash kernels: mix_state evaluations (src/hash.rs:59-86) the launches' leaf and node hashes
                           amount to -- 9 per 8-byte leaf, 10 per node (src/hash.rs:14-27,41-46); 0 elsewhere */
} smi_kernel_time;
int smi_ctx_profile(smi_ctx *ctx, int enable);
int smi_ctx_profile_read(smi_ctx *ctx, smi_kernel_time *out, size_t cap, size_t *n);
/* Restrict the brackets to launches whose kernel name contains name_part (NULL or "": all launches).  Bracketing every
 * launch perturbs what it measures -- kernels no longer run back to back and the chip clocks higher -- so bench.py
 * times its dominant kernel with only that kernel bracketed, inside an otherwise undisturbed loop. */
int smi_ctx_profile_only(smi_ctx *ctx, const char *name_part);
/* Measurement aid for the roofline leg: while enabled, every NTT pass launches its copy-only twin
 * (same tiles, same global loads and store addresses, no arithmetic) so that smi_ctx_profile
 * times what HBM delivers for each pass's access pattern.  Outputs are meaningless while it is
 * on; never enable it in product use. */
int smi_ctx_copy_probe(smi_ctx *ctx, int enable);
/* Measurement aid for the prove's roofline (bench.py `prove_roofline`): the hash's permutation and nothing else --
 * `mixes` back-to-back mix_state evaluations (src/hash.rs:59-86) per hash, two hashes per lane in the layout the
 * Merkle kernels use, enough workgroups to fill the chip, no memory traffic -- timed with HIP events on the
 * context's stream.  *mixes_per_s is the integer-VALU ceiling of this formulation of the hash on this GPU, at the
 * clock the chip sustains under that load; the Merkle kernels are reported against it. */
int smi_ctx_mix_probe(smi_ctx *ctx, uint32_t mixes, double *mixes_per_s);
/* smi_dev_lde / smi_lde at 2^20..2^22 rows: run the extension in two passes over its outputs (coset-split
 * pass A, interleaving pass B -- csrc/lde_core.h) instead of the generic three.  Same results; off by
 * default (within 2 % of the generic path on MI355X); SMI_LDE_TWO_PASS=1 sets the default. */
int smi_ctx_lde_two_pass(smi_ctx *ctx, int enable);
uint64_t smi_ctx_modulus(const smi_ctx *ctx);
uint32_t smi_ctx_two_adicity(const smi_ctx *ctx);

/* ---- field scalars (host, exact restatements needed by callers) ------------------ */
/* FiniteField::prim_nth_root (src/ff.rs:215-223): g^((p-1)/n). */
int smi_prim_nth_root(const smi_ctx *ctx, uint64_t n, uint64_t *out);
/* FiniteField::inv via Fermat; SMI_ERR_NO_INVERSE for 0 (src/ff.rs:169-178). */
int smi_ff_inv(const smi_ctx *ctx, uint64_t x, uint64_t *out);
int smi_ff_exp(const smi_ctx *ctx, uint64_t base, uint64_t e, uint64_t *out); /* src/ff.rs:200-213 */
int smi_ff_mul(const smi_ctx *ctx, uint64_t a, uint64_t b, uint64_t *out);    /* src/ff.rs:138-144 */

/* ---- univariate: host-buffer entry points ---------------------------------------- */
/* Polynomial::interpolate_domain (src/univariate/interpolate.rs:6-44) for the geometric
 * domain d[k] = offset * omega_n^k, n = 2^log_n, omega_n = prim_nth_root(n):
 * coeffs[j] = offset^-j n^-1 sum_k values[k] omega_n^-jk.  Always writes n coefficients
 * (trailing zeros included); Polynomial equality ignores them (src/univariate/mod.rs:13-39). */
int smi_intt(smi_ctx *ctx, const uint64_t *values, uint64_t *coeffs, uint32_t log_n, uint64_t offset);
/* Polynomial::eval_domain (src/univariate/eval.rs:16-21) on d[k] = offset * omega_N^k,
 * N = 2^log_N, n_coeffs <= N: evals[k] = sum_j coeffs[j] d[k]^j, in domain order. */
int smi_coset_ntt(smi_ctx *ctx, const uint64_t *coeffs, size_t n_coeffs, uint64_t *evals, uint32_t log_N,
                  uint64_t offset);
/* Polynomial::scale (src/univariate/mod.rs:99-113): out[i] = coeffs[i] * factor^i. */
int smi_poly_scale(smi_ctx *ctx, const uint64_t *coeffs, size_t n, uint64_t factor, uint64_t *out);
/* Polynomial::mul (src/univariate/mul.rs:6-29) by NTT: out gets na+nb-1 coefficients (*n_out),
 * or *n_out = 0 when either operand is the zero polynomial, as the reference returns `vec![]`. */
int smi_poly_mul(smi_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out, size_t *n_out);
/* Polynomial::div (src/univariate/div.rs:6-42): quotient and remainder of a / b, via the power-series
 * inverse of the reversed divisor (NTT products) instead of the reference's O(n*m) subtraction loop.
 * q gets deg a - deg b + 1 coefficients (*nq), r gets deg b coefficients (*nr; the reference's
 * remainder vector may carry further trailing zeros -- Polynomial equality ignores them).  When
 * deg a < deg b: *nq = 0 and r = a unchanged (na coefficients), as div.rs:10-18.  A zero divisor
 * gives SMI_ERR_POLY_DIV_BY_ZERO.  q needs room for na, r for max(na, nb) coefficients. */
int smi_poly_div(smi_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *q, size_t *nq, uint64_t *r,
                 size_t *nr);
/* Checks in O(n) whether domain[k] == domain[0]*omega_n^k (the fast-path contract of
 * interpolate_domain / eval_domain); returns SMI_OK and *offset = domain[0], or
 * SMI_ERR_NOT_GEOMETRIC. */
int smi_domain_is_geometric(const smi_ctx *ctx, const uint64_t *domain, size_t n, uint64_t *offset);

/* ---- trace: Trace::get_col / to_field_elements (src/trace.rs:21-34) --------------- */
/* Low-degree extension of a column-major trace: for each of n_cols columns (n = 2^log_n
 * values on the subgroup domain trace_offset*omega_n^k) interpolate, then evaluate on
 * lde_offset*omega_N^k, N = n << log_blowup.  out is column-major n_cols x N. */
int smi_lde(smi_ctx *ctx, const uint64_t *cols, uint32_t n_cols, uint32_t log_n, uint32_t log_blowup,
            uint64_t trace_offset, uint64_t lde_offset, uint64_t *out);
/* Row-major i128 trace (src/trace.rs:4-7, each value as 16 LE bytes) -> column-major u64,
 * `e as u64` then reduced mod p for the device (precondition H6 made explicit). */
int smi_trace_pack(const smi_ctx *ctx, const void *rows_i128, size_t n_rows, size_t n_cols, uint64_t *cols_out);

/* ---- hash / merkle ---------------------------------------------------------------- */
/* Hash::from_field_elements(&[e]) for each e (src/hash.rs:32-35 as used at src/fri.rs:118-121):
 * digests gets n x 32 bytes. */
int smi_hash_leaves(smi_ctx *ctx, const uint64_t *elems, size_t n, uint8_t *digests);
/* Hash::combine (src/hash.rs:41-46) for n pairs: out[i] = H(left_right[2i] || left_right[2i+1]). */
int smi_hash_combine_pairs(smi_ctx *ctx, const uint8_t *digests, size_t n_pairs, uint8_t *out);
/* Hash::from_bytes (src/hash.rs:7-30) of one message (device single-lane kernel). */
int smi_hash_bytes(smi_ctx *ctx, const uint8_t *msg, size_t len, uint8_t out[32]);
/* The same for n messages of msg_len bytes each (msgs: n x msg_len, out: n x 32), one device lane
 * per message: Fri::sample_indices hashes seed || counter for a run of counters (src/fri.rs:176-213). */
int smi_hash_bytes_batch(smi_ctx *ctx, const uint8_t *msgs, size_t n, size_t msg_len, uint8_t *out);
/* MerkleTree::commit (src/merkle.rs:44-65). */
int smi_merkle_commit(smi_ctx *ctx, const uint8_t *leaves, size_t n, uint8_t root[32]);
/* MerkleTree::new (src/merkle.rs:11-38); the tree (all levels) stays on the device. */
int smi_merkle_new(smi_ctx *ctx, const uint8_t *leaves, size_t n, smi_tree **out);
/* Fused leaf hashing + tree over a codeword, one element per leaf (src/fri.rs:118-127). */
int smi_merkle_from_codeword(smi_ctx *ctx, const uint64_t *codeword, size_t n, smi_tree **out);
int smi_merkle_root(smi_ctx *ctx, const smi_tree *t, uint8_t root[32]);          /* get_root, src/merkle.rs:40-42 */
/* MerkleTree::open (src/merkle.rs:67-80): path gets *depth = log2(n) digests. */
int smi_merkle_open(smi_ctx *ctx, const smi_tree *t, size_t index, uint8_t *path, size_t *depth);
/* Copies level `level` (0 = leaves) to the host: nodes[level] of src/merkle.rs:6. */
int smi_merkle_level(smi_ctx *ctx, const smi_tree *t, uint32_t level, uint8_t *out, size_t *n_out);
/* MerkleTree::verify (src/merkle.rs:82-96) for k (leaf, index, path) triples that share one depth
 * and one root -- the verifier's hot loop (src/fri.rs:464-497); ok[i] = 1 if path i authenticates. */
int smi_merkle_verify_batch(smi_ctx *ctx, const uint8_t *leaves, const uint64_t *indices, const uint8_t *paths, size_t k,
                            size_t depth, const uint8_t root[32], uint8_t *ok);
size_t smi_merkle_num_leaves(const smi_tree *t);
void smi_merkle_free(smi_tree *t);

/* ---- fri --------------------------------------------------------------------------- */
/* Fri::new's fields (src/fri.rs:8-15,30-55). */
typedef struct {
    uint64_t omega;
    uint64_t offset;
    uint64_t domain_length;
    uint64_t expansion_factor;
    uint64_t num_colinearity_tests;
} smi_fri_cfg;

int smi_fri_check(const smi_ctx *ctx, const smi_fri_cfg *cfg);                   /* asserts of src/fri.rs:37-45 */
int smi_fri_num_rounds(const smi_fri_cfg *cfg, uint64_t *rounds);                /* src/fri.rs:93-103 */
/* Fri::fold_codeword (src/fri.rs:57-91); alpha may be an unreduced u64 (src/fiat_shamir.rs:23-24). */
int smi_fri_fold(smi_ctx *ctx, const uint64_t *codeword, size_t len, uint64_t alpha, uint64_t offset,
                 uint64_t omega, uint64_t *out);
/* Fri::commit (src/fri.rs:105-156) with a fresh FiatShamir: roots gets R x 32 bytes, alphas
 * R-1 unreduced challenges, last_codeword domain_length >> (R-1) values.  *run (optional)
 * keeps every round's codeword and tree on the device (smi_fri_run_* accessors below). */
int smi_fri_commit(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint64_t *codeword, size_t len, uint8_t *roots,
                   uint64_t *alphas, uint64_t *last_codeword, size_t *last_len, smi_fri_run **run);
/* Fri::prove (src/fri.rs:250-311) with a fresh FiatShamir and ProofStream, returning
 * ProofStream::serialize (src/stream.rs:35-64).  *proof is malloc'd: release with smi_free.
 * top_indices gets num_colinearity_tests entries (the method's return value). */
int smi_fri_prove(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint64_t *codeword, size_t len, uint8_t **proof,
                  size_t *proof_len, uint64_t *top_indices);
/* The `codewords` Fri::commit returns (src/fri.rs:153-155): their count, one of them (out may be
 * NULL to query *len), and MerkleTree::open on a round's retained tree. */
/* Fri::verify (src/fri.rs:313-504) of a serialized ProofStream against a fresh FiatShamir: *accept = 1 where the
 * reference returns true, 0 where it prints a reason and returns false (smi_last_error has the reason); a
 * reference panic (e.g. a last codeword whose length is not a power of two, src/merkle.rs:13-16) is that panic's
 * status.  pv_indices / pv_values (optional, 2*t entries each) receive the (index, value) pairs the reference
 * pushes to polynomial_values.  Leaf hashes and authentication paths are checked in device batches, the last
 * layer's degree by an inverse + forward NTT; SMI_ERR_NOT_GEOMETRIC if cfg's omega does not generate the domain.
 * Two deliberate differences from the reference on malformed proofs, both failing closed (a status, never accept):
 *   - a last codeword whose length is not domain_length >> (rounds - 1) but whose Merkle root matches: the reference
 *     runs its O(L^3) Lagrange interpolation over the (then repeating or truncated) point list and returns whatever
 *     that gives or panics with "no inverse"; the NTT needs the points to be a whole coset, so this returns
 *     SMI_ERR_NOT_GEOMETRIC;
 *   - unreduced values (>= p) in a triple: the colinearity test uses the reference's own `(p + l - r) % p` in u128
 *     (src/ff.rs:154-160) including its release-build wrap for r > p + l (a debug build of the reference panics
 *     there), and the leaf is hashed from the raw u64, as in the reference. */
int smi_fri_verify(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint8_t *proof, size_t proof_len, int *accept, uint64_t *pv_indices,
                   uint64_t *pv_values, size_t *n_pv);
int smi_fri_run_num_codewords(const smi_fri_run *run, size_t *n);
int smi_fri_run_codeword(smi_fri_run *run, size_t round, uint64_t *out, size_t *len);
int smi_fri_run_open(smi_fri_run *run, size_t round, size_t index, uint8_t *path, size_t *depth);
void smi_fri_run_free(smi_fri_run *run);
void smi_free(void *p);

/* ---- device-resident entry points (u32 residues, context stream, no sync) ---------- */
int smi_dev_alloc(smi_ctx *ctx, size_t bytes, void **d_ptr);
int smi_dev_free(smi_ctx *ctx, void *d_ptr);
/* u64 host -> u32 device (checks canonical unless reduce != 0) and back. */
int smi_dev_upload_u64(smi_ctx *ctx, const uint64_t *host, size_t n, uint32_t *d_out, int reduce);
int smi_dev_download_u64(smi_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *host);

/* Batched NTT kernel driver.  For each of `batch` columns (column c at d_in + c*in_stride,
 * d_out + c*out_stride, strides in elements):
 *   inverse == 0: out[k] = sum_{j<n_in} in[j] (offset*omega_N^k)^j        (coset NTT, zero-padded)
 *   inverse != 0: out[j] = scale * offset^-j N^-1 sum_k in[k] omega_N^-jk (n_in must equal N)
 * N = 2^log_n; natural order in and out; d_in may equal d_out (columns must not overlap).
 * post_scale (canonical, 1 = none) multiplies every output of the inverse transform by
 * post_scale^j -- the fused `Polynomial::scale` (src/univariate/mod.rs:99-113) of an LDE. */
int smi_dev_ntt(smi_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint32_t log_n, size_t n_in, uint32_t batch,
                size_t in_stride, size_t out_stride, int inverse, uint64_t offset, uint64_t post_scale);
/* LDE of a column-major device trace: d_out is n_cols x (n << log_blowup), stride N. */
int smi_dev_lde(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, uint32_t log_n, uint32_t log_blowup,
                uint64_t trace_offset, uint64_t lde_offset, uint32_t *d_out);
/* Leaf digests only / fused leaf hashing + all tree levels.  d_nodes: (2n-1) x 32 bytes,
 * level 0 (leaf digests) first, root last -- `nodes` of src/merkle.rs:18-33 back to back. */
int smi_dev_hash_leaves(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_digests);
int smi_dev_merkle_build(smi_ctx *ctx, const uint32_t *d_elems, size_t n, uint8_t *d_nodes);
/* Row-leaf tree (build-defined leaf rule): leaf i = Hash::from_field_elements of row i
 * of n_cols columns (column c at d_cols + c*col_stride), i.e. src/hash.rs:32-35 applied to the row
 * instead of to a single element; with 4 columns a leaf is one 32-byte chunk and costs what a
 * single-element leaf costs, so one tree replaces four. */
int smi_dev_merkle_build_rows(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t col_stride, size_t n, uint8_t *d_nodes);
/* Tree over precomputed 32-byte leaves already at d_nodes[0 .. n*32). */
int smi_dev_merkle_from_digests(smi_ctx *ctx, size_t n, uint8_t *d_nodes);
/* Hash::from_bytes (src/hash.rs:7-30) of a message in device memory, digest to device memory: a
 * transcript that lives on the device (roots in, the challenge = first 8 digest bytes out, see
 * src/fiat_shamir.rs:19-25) needs no host round trip per round. */
int smi_dev_hash_bytes(smi_ctx *ctx, const uint8_t *d_msg, size_t len, uint8_t *d_out32);
/* Fri::fold_codeword with alpha read from device memory (*d_alpha: one unreduced u64). */
int smi_dev_fri_fold(smi_ctx *ctx, const uint32_t *d_in, size_t len, const uint64_t *d_alpha, uint64_t offset,
                     uint64_t omega, uint32_t *d_out);
/* The same fold for one shard of a codeword distributed over several GPUs: `count`
 * outputs starting at global output index index0, d_lo[k] = c[index0+k], d_hi[k] =
 * c[index0+k+full_len/2] (the second operand arrives from the partner GPU). */
int smi_dev_fri_fold_shard(smi_ctx *ctx, const uint32_t *d_lo, const uint32_t *d_hi, size_t count, size_t index0,
                           size_t full_len, const uint64_t *d_alpha, uint64_t offset, uint64_t omega, uint32_t *d_out);
/* Fri::commit + the query phase of Fri::prove over a device codeword.  Roots, alphas and
 * the serialized proof are produced without a host round trip per round: Fiat-Shamir and
 * index sampling run in single-lane device kernels. */
int smi_dev_fri_prove(smi_ctx *ctx, const smi_fri_cfg *cfg, const uint32_t *d_codeword, size_t len, uint8_t **proof,
                      size_t *proof_len, uint64_t *top_indices, smi_fri_run **run);

/* out[i] = sum_c (weights[c] mod p) * cols[c*stride + i]; d_weights holds n_cols unreduced u64
 * challenges on the device (random linear combination of committed columns; build-defined,
 * the reference has no prover above Fri::prove -- SURVEY F5). */
int smi_dev_combine_columns(smi_ctx *ctx, const uint32_t *d_cols, uint32_t n_cols, size_t len, size_t stride,
                            const uint64_t *d_weights, uint32_t *d_out);
/* Build-defined composition of the reference primitives: column-major device
 * trace -> LDE (blowup 2^log_blowup on the coset lde_offset*<w_N>) -> one Merkle tree per column
 * (one element per leaf, src/fri.rs:118-121) -> fresh FiatShamir absorbs the column roots and
 * draws one weight per column -> Fri::prove (src/fri.rs:250-311, expansion_factor = blowup) on
 * the weighted sum.  column_roots (host, optional) gets n_cols x 32 bytes; *proof is the
 * serialized FRI ProofStream (smi_free); stage_ms (optional) gets the HIP-event times of
 * {lde, column commits, combine, fri} in milliseconds. */
typedef struct {
    uint32_t log_n, log_blowup, n_cols;
    uint32_t row_leaves;   /* 0: one tree per column, one element per leaf (the reference's leaf rule, src/fri.rs:118-121);
                              1: one tree over the rows (smi_dev_merkle_build_rows) -- a build-defined variant */
    uint64_t trace_offset, lde_offset, num_colinearity_tests;
    uint64_t open_columns; /* 1: after the FRI objects, bind the combined codeword to the committed columns -- for every
                              colinearity test s, with a = top_index[s] mod N/2 and b = a + N/2 (the layer-0 positions
                              Fri::query opens, src/fri.rs:215-248): FieldElements([col_0[a] .. col_{W-1}[a]]),
                              FieldElements([col_0[b] .. col_{W-1}[b]]) for s = 0 .. t-1, then MerklePath(col_c, a),
                              MerklePath(col_c, b) for every s and, inside it, every c (tags and widths of src/stream.rs:35-64).  A verifier checks each path against
                              column root c and sum_c weight_c * col_c[a] against the FRI triple's value.  Column trees only. */
} smi_stark_cfg;
int smi_dev_stark_prove(smi_ctx *ctx, const smi_stark_cfg *cfg, const uint32_t *d_trace_cols, uint8_t *column_roots,
                        uint8_t **proof, size_t *proof_len, uint64_t *top_indices, double *stage_ms);

/* Verifier of smi_dev_stark_prove / smi_mgpu_stark_prove (column trees, cfg->open_columns != 0): Fri::verify of the
 * leading objects on the domain lde_offset * <w_N>, then the column openings -- every authentication path against
 * its column root (column_roots: n_cols x 32) and sum_c weight_c * col_c[a] against the layer-0 triple -- with
 * the weights re-derived from the column roots.  *accept as in smi_fri_verify.  A proof made with open_columns == 0
 * is exactly Fri::prove's bytes: no object in it refers to the column roots, so accepting it here would present a
 * low-degree proof of an unrelated codeword as a proof about these columns.  The call therefore returns
 * SMI_ERR_COLUMNS_NOT_BOUND (*accept = 0, nothing verified) when cfg->open_columns == 0; check such a proof
 * with smi_fri_verify. */
int smi_stark_verify(smi_ctx *ctx, const smi_stark_cfg *cfg, const uint8_t *column_roots, const uint8_t *proof, size_t proof_len,
                     int *accept);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------
 * Fri::commit / Fri::prove (src/fri.rs:105-156, 250-311) over ONE codeword sharded in contiguous
 * blocks (rank g holds [g*N/G, (g+1)*N/G)), and the build-defined trace -> proof composition of
 * smi_dev_stark_prove over the G ranks.  Every rank makes the same calls in the same order; the
 * collectives (all-gather of the G sub-roots per tree, one grouped send/recv per fold and for the
 * extension's all-to-all, one byte-sum all-reduce of the proof) run inside the library on the
 * context's stream.  Results are bit-identical to the single-GPU entry points by construction and
 * as far as they have been run: at every world size (2, 4, 8) on the CPU instantiation of the same loop,
 * with 2 and 4 ranks on one GPU through smi_mgpu_create_with (host-staged collectives), and over a real
 * RCCL communicator at world size 1.  An RCCL communicator with MORE THAN ONE RANK has not been exercised:
 * the development boxes have one GPU.  bench.py therefore self-checks three entry points against their
 * single-GPU twins on every rank before it times anything on several GPUs.  World sizes are powers of two. */
typedef struct smi_mgpu smi_mgpu;
#define SMI_MGPU_ID_BYTES 128
/* rank 0: ncclGetUniqueId; the caller carries the 128 bytes to the other ranks (any channel) */
int smi_mgpu_unique_id(uint8_t id[SMI_MGPU_ID_BYTES]);
/* ncclCommInitRank on the context's device; collective over all ranks */
int smi_mgpu_create(smi_ctx *ctx, const uint8_t id[SMI_MGPU_ID_BYTES], int rank, int world, smi_mgpu **out);
/* The same prover over caller-supplied collectives (tests without several GPUs, other transports).
 * Pointers are device memory; the library drains its stream before each call and expects the
 * operation to have completed on return (0 = ok).  exchange: the k-th send to a peer matches that
 * peer's k-th recv from this rank. */
typedef struct {
    void *user;
    int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank);
    int (*exchange)(void *user, int n_send, const int *send_peer, void *const *d_send, const size_t *send_bytes, int n_recv,
                    const int *recv_peer, void *const *d_recv, const size_t *recv_bytes);
    int (*all_reduce_sum_u8)(void *user, void *d_buf, size_t bytes);
} smi_mgpu_coll;
int smi_mgpu_create_with(smi_ctx *ctx, const smi_mgpu_coll *ops, int rank, int world, smi_mgpu **out);
void smi_mgpu_destroy(smi_mgpu *m);
/* blocks shorter than this are all-gathered and the remaining rounds run replicated (default 2^18) */
int smi_mgpu_set_min_block(smi_mgpu *m, size_t min_block);
/* Fri::commit: d_block = this rank's block of the initial codeword (device u32).  roots: R x 32,
 * alphas: R-1 unreduced u64, last codeword as u64 -- on every rank. */
int smi_mgpu_fri_commit(smi_mgpu *m, const smi_fri_cfg *cfg, const uint32_t *d_block, size_t block_len, uint8_t *roots, uint64_t *alphas,
                        uint64_t *last_codeword, size_t *last_len);
/* Fri::prove: the serialized ProofStream (smi_free) and the top-level indices on every rank. */
int smi_mgpu_fri_prove(smi_mgpu *m, const smi_fri_cfg *cfg, const uint32_t *d_block, size_t block_len, uint8_t **proof, size_t *proof_len,
                       uint64_t *top_indices);
/* The extension of smi_dev_lde sharded by (column, coset) units (n_cols << log_blowup and 2^log_n
 * multiples of the world size, log_blowup <= 4): d_trace_cols = the whole trace on every rank;
 * d_out_blocks gets this rank's natural-order block of every column (n_cols x N/G, stride N/G). */
int smi_mgpu_lde(smi_mgpu *m, const uint32_t *d_trace_cols, uint32_t n_cols, uint32_t log_n, uint32_t log_blowup, uint64_t trace_offset,
                 uint64_t lde_offset, uint32_t *d_out_blocks);
/* ONE transform of 2^log_n points over the G ranks (BASELINE configs[3]; the reference has no NTT, this is
 * Polynomial::eval_domain / interpolate_domain on a geometric domain of that size, src/univariate/eval.rs:16-21,
 * interpolate.rs:6-44) on the ordinary pass pipeline with ONE all-to-all.  With R_0 = 2^*log_r0 the
 * plan's first digit (smi_mgpu_ntt_first_digit) and B = N / R_0:
 *   d_strip: this rank's columns [rank*B/G, (rank+1)*B/G) of the row-major [R_0][B] view of the input, as
 *            [R_0][B/G] (clobbered);
 *   d_out  : N/G outputs, X[k_0 + R_0*rest] at rest*(R_0/G) + (k_0 - rank*R_0/G) -- natural-order runs of R_0/G.
 * Forward: evaluations on offset*<w_N>; inverse (offset must be 1): coefficients from values.  At G = 1 it is
 * the direct transform. */
int smi_mgpu_ntt(smi_mgpu *m, uint32_t *d_strip, uint32_t *d_out, uint32_t log_n, int inverse, uint64_t offset);
/* The same transform with the result as this rank's CONTIGUOUS natural-order block: d_out gets X[rank*N/G ..
 * (rank+1)*N/G), the layout smi_mgpu_fri_commit / smi_mgpu_fri_prove take (so a 2^26-point evaluation can be
 * committed without leaving the GPUs).  Costs a second all-to-all of the same volume: with one exchange the rank
 * that owns k_0 holds every R_0-th output, and a contiguous output range is a range of the other index. */
int smi_mgpu_ntt_natural(smi_mgpu *m, uint32_t *d_strip, uint32_t *d_out, uint32_t log_n, int inverse, uint64_t offset);
int smi_mgpu_ntt_first_digit(uint32_t log_n, uint32_t *log_r0);
/* smi_dev_stark_prove (column trees) over the G ranks: same column roots, same proof bytes. */
int smi_mgpu_stark_prove(smi_mgpu *m, const smi_stark_cfg *cfg, const uint32_t *d_trace_cols, uint8_t *column_roots, uint8_t **proof,
                         size_t *proof_len, uint64_t *top_indices);

#ifdef __cplusplus
}
#endif
#endif /* STARK_MI_H */
