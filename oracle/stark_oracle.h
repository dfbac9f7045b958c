/*
 * stark_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Op-for-op C restatement of the hot path of 0xSooki/stark-rs (reference snapshot
 * 2026-01-02): prime-field arithmetic, Lagrange interpolation, power-sum evaluation,
 * the byte hash, the Merkle tree, Fiat-Shamir, the proof stream and FRI
 * prove/verify.  Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product (libstarkmi.so) never links,
 * loads or calls anything in oracle/.
 *
 * Parity pin status: field / interpolation / evaluation / scale / FRI-accept results
 * are pinned by the reference's own known-answer tests (tests/test_oracle_kats.py
 * transliterates them).  Digests, Merkle roots, Fiat-Shamir challenges, sampled
 * indices and serialized proof bytes are NOT pinned by any reference vector
 * (hash.rs:106-149 assert structure only) -- "parity unpinned" for
 * those values; they rest on restatement fidelity plus prove->verify round trips.
 *
 * The reference cannot be compiled here (Rust; no cargo/rustc in the image).
 */
#ifndef STARK_ORACLE_H
#define STARK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO_P_REF 998244353ULL /* src/ff.rs:192 */

/* ---- error channel: the reference panics; the oracle records the message ---- */
/* Returns the last recorded panic message ("" if none) and clears nothing. */
const char *so_last_panic(void);
void so_clear_panic(void);

/* ---- field (src/ff.rs:108-233), p passed explicitly (FiniteField{p}) ---- */
uint64_t so_ff_add(uint64_t p, uint64_t l, uint64_t r);   /* ff.rs:146-152 */
uint64_t so_ff_sub(uint64_t p, uint64_t l, uint64_t r);   /* ff.rs:154-160 */
uint64_t so_ff_mul(uint64_t p, uint64_t l, uint64_t r);   /* ff.rs:138-144 */
uint64_t so_ff_neg(uint64_t p, uint64_t x);               /* ff.rs:162-167 */
uint64_t so_ff_inv(uint64_t p, uint64_t x);               /* ff.rs:169-178 (panics "no inverse") */
uint64_t so_ff_div(uint64_t p, uint64_t l, uint64_t r);   /* ff.rs:181-189 */
uint64_t so_ff_exp(uint64_t p, uint64_t base, uint64_t e);/* ff.rs:200-213 */
uint64_t so_ff_g(uint64_t p);                             /* ff.rs:191-197 (asserts p==998244353) */
uint64_t so_ff_prim_nth_root(uint64_t p, uint64_t n);     /* ff.rs:215-223 */
/* p-generic variant used for the second prime: g^((p-1)/n). */
uint64_t so_ff_prim_nth_root_g(uint64_t p, uint64_t g, uint64_t n);
uint64_t so_ff_sample(uint64_t p, const uint8_t *salt, size_t len); /* ff.rs:225-232 */
/* utils.rs:3-13; results are i128 in the reference, here split lo/hi is not needed
 * for p < 2^63: returned as int64 triples via out[3] (gcd, x, y) truncated. */
void so_xgcd(uint64_t x, uint64_t y, int64_t out[3]);

/* ---- polynomials (src/univariate/*.rs): ascending coefficients, u64 values ---- */
int64_t so_poly_deg(const uint64_t *c, size_t n);                     /* mod.rs:54-68 */
int so_poly_eq(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb); /* mod.rs:13-39 */
/* Each returns the length written to out (caller sizes out generously). */
size_t so_poly_add(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out); /* add.rs:6-32 */
size_t so_poly_sub(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out); /* sub.rs:8-34 */
size_t so_poly_mul(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out); /* mul.rs:6-29 */
size_t so_poly_scale(uint64_t p, const uint64_t *a, size_t na, uint64_t factor, uint64_t *out);           /* mod.rs:99-113 */
uint64_t so_poly_eval(uint64_t p, const uint64_t *c, size_t n, uint64_t x);                               /* eval.rs:6-14 */
void so_poly_eval_domain(uint64_t p, const uint64_t *c, size_t n, const uint64_t *dom, size_t nd, uint64_t *out); /* eval.rs:16-21 */
/* interpolate.rs:6-44.  out must hold n entries; returns the length of the result:
 * n when any value is nonzero, 0 when every value is zero and n > 1 (mul.rs:7-12 zero
 * short-circuit + add.rs:7-12), 1 when n == 1 -- see H8 in SURVEY.md. */
size_t so_poly_interpolate_domain(uint64_t p, const uint64_t *dom, const uint64_t *vals, size_t n, uint64_t *out);
size_t so_poly_zerofier(uint64_t p, const uint64_t *dom, size_t n, uint64_t *out);                        /* mod.rs:77-96 */
/* div.rs:6-41: q_out/r_out sized >= na; lengths via *nq,*nr */
void so_poly_div(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                 uint64_t *q_out, size_t *nq, uint64_t *r_out, size_t *nr);
size_t so_poly_exp(uint64_t p, const uint64_t *a, size_t na, uint64_t e, uint64_t *out, size_t out_cap); /* exp.rs:6-33 */
int so_poly_test_colinearity(uint64_t p, const uint64_t *xs, const uint64_t *ys, size_t n);              /* mod.rs:145-152 */

/* ---- hash (src/hash.rs) ---- */
void so_hash_from_bytes(const uint8_t *bytes, size_t len, uint8_t out[32]);          /* hash.rs:7-30 */
void so_hash_from_field_elements(const uint64_t *e, size_t n, uint8_t out[32]);      /* hash.rs:32-35 */
void so_hash_from_u64(uint64_t v, uint8_t out[32]);                                  /* hash.rs:37-39 */
void so_hash_combine(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]);     /* hash.rs:41-46 */

/* ---- merkle (src/merkle.rs) ---- */
/* nodes_out holds all levels back to back: level 0 (n digests) | level 1 (n/2) | ... | root;
 * 2n-1 digests in all (merkle.rs:18-33 `nodes`).  Returns 0, or -1 on the panics at :12-16. */
int so_merkle_new(const uint8_t *leaves, size_t n, uint8_t *nodes_out);
int so_merkle_commit(const uint8_t *leaves, size_t n, uint8_t root[32]);             /* merkle.rs:44-65 */
/* merkle.rs:67-80: path_out gets log2(n) digests; returns depth or -1 ("Index out of bounds"). */
int so_merkle_open(const uint8_t *nodes, size_t n, size_t index, uint8_t *path_out);
int so_merkle_verify(const uint8_t leaf[32], size_t index, const uint8_t *path, size_t depth,
                     const uint8_t root[32]);                                        /* merkle.rs:82-96 */

/* ---- fiat-shamir (src/fiat_shamir.rs) ---- */
typedef struct so_fs so_fs;
so_fs *so_fs_new(void);
void so_fs_free(so_fs *);
void so_fs_absorb(so_fs *, const uint8_t *data, size_t len);   /* fiat_shamir.rs:15-17 */
uint64_t so_fs_challenge(const so_fs *);                       /* fiat_shamir.rs:19-25 (unreduced u64) */

/* ---- trace (src/trace.rs) ---- */
/* trace.rs:29-34: `e as u64` on an i128 keeps the low 64 bits; the oracle takes the
 * i128 as (lo,hi) pairs and returns lo, unreduced. */
void so_trace_to_field_elements(const uint64_t *lo, const uint64_t *hi, size_t n, uint64_t *out);
/* trace.rs:36-49 fibonacci; emits low 64 bits of each i128 row value, rows < 184. */
void so_trace_fibonacci(size_t length, uint64_t *out_lo, uint64_t *out_hi);

/* ---- FRI (src/fri.rs) ---- */
typedef struct {
    uint64_t p;
    uint64_t omega;
    uint64_t offset;
    uint64_t domain_length;
    uint64_t expansion_factor;
    uint64_t num_colinearity_tests;
} so_fri_cfg;

int so_fri_new_check(const so_fri_cfg *);                                         /* fri.rs:37-45 asserts; 0 ok */
uint64_t so_fri_num_rounds(const so_fri_cfg *);                                   /* fri.rs:93-103 */
void so_fri_fold_codeword(const so_fri_cfg *, const uint64_t *cw, size_t len, uint64_t alpha,
                          uint64_t offset, uint64_t omega, uint64_t *out);        /* fri.rs:57-91 */
void so_fri_eval_domain(const so_fri_cfg *, size_t round, uint64_t *out);         /* fri.rs:158-166 */
size_t so_fri_sample_index(const uint8_t *bytes, size_t nbytes, size_t size);     /* fri.rs:168-174 */
/* fri.rs:176-213; returns 0 or -1 on the asserts */
int so_fri_sample_indices(const uint8_t *seed, size_t seed_len, size_t size, size_t reduced_size,
                          size_t number, uint64_t *out);
/* fri.rs:250-311 + stream.rs:35-64: runs prove with a fresh FiatShamir/ProofStream and
 * returns the serialized proof.  *proof_out is malloc'd (free with so_free).
 * top_indices_out gets num_colinearity_tests entries.  Returns 0, or -1 on a panic. */
int so_fri_prove(const so_fri_cfg *, const uint64_t *codeword, size_t len,
                 uint8_t **proof_out, size_t *proof_len, uint64_t *top_indices_out);
/* Also exposes the commit-phase artefacts: roots (R x 32 B), alphas (R-1), codeword lengths. */
int so_fri_commit_trace(const so_fri_cfg *, const uint64_t *codeword, size_t len,
                        uint8_t *roots_out, uint64_t *alphas_out, uint64_t *last_codeword_out,
                        size_t *last_len);
/* fri.rs:313-504 after stream.rs:66-168 deserialize.  Returns 1 accept / 0 reject.
 * pv_idx/pv_val (optional, may be NULL) receive the 2t (index,value) top-layer pairs. */
int so_fri_verify(const so_fri_cfg *, const uint8_t *proof, size_t proof_len,
                  uint64_t *pv_idx, uint64_t *pv_val, size_t *pv_n);
const char *so_fri_last_reject(void);

void so_free(void *);

/* ---- fast CPU restatement of the NTT identities (NOT in the reference; F1) ----
 * Radix-2 transforms that must equal so_poly_interpolate_domain / so_poly_eval_domain
 * on geometric domains offset*omega^k; tests prove that for n <= 2^10 before they are
 * used as the large-size checker and as the "fair algorithmic" CPU baseline. */
void so_fast_intt(uint64_t p, uint64_t omega, uint64_t offset, const uint64_t *vals, size_t n, uint64_t *coeffs);
void so_fast_coset_ntt(uint64_t p, uint64_t omega_N, uint64_t offset, const uint64_t *coeffs, size_t nc,
                       size_t N, uint64_t *evals);
void so_fast_fold(uint64_t p, const uint64_t *cw, size_t len, uint64_t alpha, uint64_t offset,
                  uint64_t omega, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
