//! stark_mi.rs -- the Rust side of the drop-in boundary: raw declarations of every entry point of
//! `include/stark_mi.h` (libstarkmi.so, hand-written HIP for gfx950 behind a C ABI) and thin safe wrappers for
//! the methods of 0xSooki/stark-rs that stand in front of it.  Add to the reference as `src/mi.rs`
//! (`mod mi;`) with the `build.rs` of INTEGRATION.md; the reference itself has no FFI and no dependencies,
//! so this file uses `std` only.
//!
//! What each wrapper replaces (reference file:line):
//!   interpolate_domain  src/univariate/interpolate.rs:6-44     eval_domain   src/univariate/eval.rs:16-21
//!   scale               src/univariate/mod.rs:99-113           mul / div     src/univariate/mul.rs:6-29, div.rs:6-52
//!   hash_leaves         src/hash.rs:32-35 via src/fri.rs:118-121   combine_pairs  src/hash.rs:41-46
//!   MerkleTree          src/merkle.rs:11-80                    fold_codeword src/fri.rs:57-91
//!   fri_prove           src/fri.rs:250-311                     fri_verify    src/fri.rs:313-504
//!   lde / trace_pack    src/trace.rs:21-34 + per-column interpolate / evaluate (no reference function)
//!
//! Values cross the boundary as the reference's wire format: field elements as little-endian u64
//! (`FieldElement::value`, src/stream.rs:45), digests as 32 raw bytes (`Hash.0`).  A non-zero status becomes
//! the reference's own panic message (`check`), so `#[should_panic(expected = "...")]` tests keep passing.
//!
//! The image this was written in has no rustc: tests/test_rust_binding.py checks every declaration below
//! against the header (symbol set, arity, pointer depth and constness, integer widths, struct layouts).
#![allow(non_camel_case_types, dead_code, clippy::too_many_arguments, clippy::missing_safety_doc)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

// BEGIN GENERATED (tools/gen_rust_bindings.py from include/stark_mi.h) -- do not edit by hand
/// Status codes (include/stark_mi.h): 0 = ok; -1..-18 mirror a reference panic; the rest are contract or runtime errors.
pub const SMI_OK: c_int = 0;
pub const SMI_ERR_NO_INVERSE: c_int = -1;
pub const SMI_ERR_DIV_BY_ZERO: c_int = -2;
pub const SMI_ERR_NOT_POW2: c_int = -3;
pub const SMI_ERR_ROOT_TOO_LARGE: c_int = -4;
pub const SMI_ERR_EMPTY_LEAVES: c_int = -5;
pub const SMI_ERR_LEAVES_NOT_POW2: c_int = -6;
pub const SMI_ERR_INDEX_OOB: c_int = -7;
pub const SMI_ERR_DOMAIN_NOT_POW2: c_int = -8;
pub const SMI_ERR_EXPANSION_NOT_POW2: c_int = -9;
pub const SMI_ERR_EXPANSION_TOO_SMALL: c_int = -10;
pub const SMI_ERR_CODEWORD_LEN: c_int = -11;
pub const SMI_ERR_SAMPLE_ENTROPY: c_int = -12;
pub const SMI_ERR_SAMPLE_TOO_MANY: c_int = -13;
pub const SMI_ERR_LEN_MISMATCH: c_int = -14;
pub const SMI_ERR_EMPTY_DOMAIN: c_int = -15;
pub const SMI_ERR_WRONG_FIELD: c_int = -16;
pub const SMI_ERR_POLY_DIV_BY_ZERO: c_int = -18;
pub const SMI_ERR_NO_ROUNDS: c_int = -17;
pub const SMI_ERR_BAD_ARG: c_int = -50;
pub const SMI_ERR_NON_CANONICAL: c_int = -51;
pub const SMI_ERR_UNSUPPORTED_PRIME: c_int = -52;
pub const SMI_ERR_NOT_GEOMETRIC: c_int = -53;
pub const SMI_ERR_COLUMNS_NOT_BOUND: c_int = -54;
pub const SMI_ERR_HIP: c_int = -100;
pub const SMI_ERR_NO_DEVICE: c_int = -101;
pub const SMI_ERR_OOM: c_int = -102;
pub const SMI_ERR_RCCL: c_int = -103;
pub const SMI_MGPU_ID_BYTES: usize = 128;

#[repr(C)] pub struct smi_ctx { _p: [u8; 0] }
#[repr(C)] pub struct smi_tree { _p: [u8; 0] }
#[repr(C)] pub struct smi_fri_run { _p: [u8; 0] }
#[repr(C)] pub struct smi_mgpu { _p: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct smi_kernel_time {
    pub name: [c_char; 56],
    pub launches: u32,
    pub total_ms: f64,
    pub alg_bytes: f64,
    pub alg_mixes: f64,
}
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct smi_fri_cfg {
    pub omega: u64,
    pub offset: u64,
    pub domain_length: u64,
    pub expansion_factor: u64,
    pub num_colinearity_tests: u64,
}
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct smi_stark_cfg {
    pub log_n: u32,
    pub log_blowup: u32,
    pub n_cols: u32,
    pub row_leaves: u32,
    pub trace_offset: u64,
    pub lde_offset: u64,
    pub num_colinearity_tests: u64,
    pub open_columns: u64,
}

/// smi_mgpu_coll: three caller-supplied collectives (device pointers; 0 = ok), how another transport is plugged in
#[repr(C)] #[derive(Clone, Copy)]
pub struct smi_mgpu_coll {
    pub user: *mut c_void,
    pub all_gather: Option<unsafe extern "C" fn(user: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: usize) -> c_int>,
    pub exchange: Option<unsafe extern "C" fn(user: *mut c_void, n_send: c_int, send_peer: *const c_int, d_send: *const *mut c_void, send_bytes: *const usize,
                                                n_recv: c_int, recv_peer: *const c_int, d_recv: *const *mut c_void, recv_bytes: *const usize) -> c_int>,
    pub all_reduce_sum_u8: Option<unsafe extern "C" fn(user: *mut c_void, d_buf: *mut c_void, bytes: usize) -> c_int>,
}

#[link(name = "starkmi")]
extern "C" {
    pub fn smi_status_string(status: c_int) -> *const c_char;
    pub fn smi_last_error(ctx: *const smi_ctx) -> *const c_char;
    pub fn smi_version() -> *const c_char;
    pub fn smi_ctx_create(p: u64, g: u64, device: c_int, out: *mut *mut smi_ctx) -> c_int;
    pub fn smi_ctx_destroy(ctx: *mut smi_ctx);
    pub fn smi_ctx_set_stream(ctx: *mut smi_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn smi_ctx_sync(ctx: *mut smi_ctx) -> c_int;
    pub fn smi_ctx_profile(ctx: *mut smi_ctx, enable: c_int) -> c_int;
    pub fn smi_ctx_profile_read(ctx: *mut smi_ctx, out: *mut smi_kernel_time, cap: usize, n: *mut usize) -> c_int;
    pub fn smi_ctx_profile_only(ctx: *mut smi_ctx, name_part: *const c_char) -> c_int;
    pub fn smi_ctx_copy_probe(ctx: *mut smi_ctx, enable: c_int) -> c_int;
    pub fn smi_ctx_mix_probe(ctx: *mut smi_ctx, mixes: u32, mixes_per_s: *mut f64) -> c_int;
    pub fn smi_ctx_lde_two_pass(ctx: *mut smi_ctx, enable: c_int) -> c_int;
    pub fn smi_ctx_modulus(ctx: *const smi_ctx) -> u64;
    pub fn smi_ctx_two_adicity(ctx: *const smi_ctx) -> u32;
    pub fn smi_prim_nth_root(ctx: *const smi_ctx, n: u64, out: *mut u64) -> c_int;
    pub fn smi_ff_inv(ctx: *const smi_ctx, x: u64, out: *mut u64) -> c_int;
    pub fn smi_ff_exp(ctx: *const smi_ctx, base: u64, e: u64, out: *mut u64) -> c_int;
    pub fn smi_ff_mul(ctx: *const smi_ctx, a: u64, b: u64, out: *mut u64) -> c_int;
    pub fn smi_intt(ctx: *mut smi_ctx, values: *const u64, coeffs: *mut u64, log_n: u32, offset: u64) -> c_int;
    pub fn smi_coset_ntt(ctx: *mut smi_ctx, coeffs: *const u64, n_coeffs: usize, evals: *mut u64, log_N: u32, offset: u64) -> c_int;
    pub fn smi_poly_scale(ctx: *mut smi_ctx, coeffs: *const u64, n: usize, factor: u64, out: *mut u64) -> c_int;
    pub fn smi_poly_mul(ctx: *mut smi_ctx, a: *const u64, na: usize, b: *const u64, nb: usize, out: *mut u64, n_out: *mut usize) -> c_int;
    pub fn smi_poly_div(ctx: *mut smi_ctx, a: *const u64, na: usize, b: *const u64, nb: usize, q: *mut u64, nq: *mut usize, r: *mut u64, nr: *mut usize) -> c_int;
    pub fn smi_domain_is_geometric(ctx: *const smi_ctx, domain: *const u64, n: usize, offset: *mut u64) -> c_int;
    pub fn smi_lde(ctx: *mut smi_ctx, cols: *const u64, n_cols: u32, log_n: u32, log_blowup: u32, trace_offset: u64, lde_offset: u64, out: *mut u64) -> c_int;
    pub fn smi_trace_pack(ctx: *const smi_ctx, rows_i128: *const c_void, n_rows: usize, n_cols: usize, cols_out: *mut u64) -> c_int;
    pub fn smi_hash_leaves(ctx: *mut smi_ctx, elems: *const u64, n: usize, digests: *mut u8) -> c_int;
    pub fn smi_hash_combine_pairs(ctx: *mut smi_ctx, digests: *const u8, n_pairs: usize, out: *mut u8) -> c_int;
    pub fn smi_hash_bytes(ctx: *mut smi_ctx, msg: *const u8, len: usize, out: *mut u8) -> c_int;
    pub fn smi_hash_bytes_batch(ctx: *mut smi_ctx, msgs: *const u8, n: usize, msg_len: usize, out: *mut u8) -> c_int;
    pub fn smi_merkle_commit(ctx: *mut smi_ctx, leaves: *const u8, n: usize, root: *mut u8) -> c_int;
    pub fn smi_merkle_new(ctx: *mut smi_ctx, leaves: *const u8, n: usize, out: *mut *mut smi_tree) -> c_int;
    pub fn smi_merkle_from_codeword(ctx: *mut smi_ctx, codeword: *const u64, n: usize, out: *mut *mut smi_tree) -> c_int;
    pub fn smi_merkle_root(ctx: *mut smi_ctx, t: *const smi_tree, root: *mut u8) -> c_int;
    pub fn smi_merkle_open(ctx: *mut smi_ctx, t: *const smi_tree, index: usize, path: *mut u8, depth: *mut usize) -> c_int;
    pub fn smi_merkle_level(ctx: *mut smi_ctx, t: *const smi_tree, level: u32, out: *mut u8, n_out: *mut usize) -> c_int;
    pub fn smi_merkle_verify_batch(ctx: *mut smi_ctx, leaves: *const u8, indices: *const u64, paths: *const u8, k: usize, depth: usize, root: *const u8, ok: *mut u8) -> c_int;
    pub fn smi_merkle_num_leaves(t: *const smi_tree) -> usize;
    pub fn smi_merkle_free(t: *mut smi_tree);
    pub fn smi_fri_check(ctx: *const smi_ctx, cfg: *const smi_fri_cfg) -> c_int;
    pub fn smi_fri_num_rounds(cfg: *const smi_fri_cfg, rounds: *mut u64) -> c_int;
    pub fn smi_fri_fold(ctx: *mut smi_ctx, codeword: *const u64, len: usize, alpha: u64, offset: u64, omega: u64, out: *mut u64) -> c_int;
    pub fn smi_fri_commit(ctx: *mut smi_ctx, cfg: *const smi_fri_cfg, codeword: *const u64, len: usize, roots: *mut u8, alphas: *mut u64, last_codeword: *mut u64, last_len: *mut usize, run: *mut *mut smi_fri_run) -> c_int;
    pub fn smi_fri_prove(ctx: *mut smi_ctx, cfg: *const smi_fri_cfg, codeword: *const u64, len: usize, proof: *mut *mut u8, proof_len: *mut usize, top_indices: *mut u64) -> c_int;
    pub fn smi_fri_verify(ctx: *mut smi_ctx, cfg: *const smi_fri_cfg, proof: *const u8, proof_len: usize, accept: *mut c_int, pv_indices: *mut u64, pv_values: *mut u64, n_pv: *mut usize) -> c_int;
    pub fn smi_fri_run_num_codewords(run: *const smi_fri_run, n: *mut usize) -> c_int;
    pub fn smi_fri_run_codeword(run: *mut smi_fri_run, round: usize, out: *mut u64, len: *mut usize) -> c_int;
    pub fn smi_fri_run_open(run: *mut smi_fri_run, round: usize, index: usize, path: *mut u8, depth: *mut usize) -> c_int;
    pub fn smi_fri_run_free(run: *mut smi_fri_run);
    pub fn smi_free(p: *mut c_void);
    pub fn smi_dev_alloc(ctx: *mut smi_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn smi_dev_free(ctx: *mut smi_ctx, d_ptr: *mut c_void) -> c_int;
    pub fn smi_dev_upload_u64(ctx: *mut smi_ctx, host: *const u64, n: usize, d_out: *mut u32, reduce: c_int) -> c_int;
    pub fn smi_dev_download_u64(ctx: *mut smi_ctx, d_in: *const u32, n: usize, host: *mut u64) -> c_int;
    pub fn smi_dev_ntt(ctx: *mut smi_ctx, d_in: *const u32, d_out: *mut u32, log_n: u32, n_in: usize, batch: u32, in_stride: usize, out_stride: usize, inverse: c_int, offset: u64, post_scale: u64) -> c_int;
    pub fn smi_dev_lde(ctx: *mut smi_ctx, d_cols: *const u32, n_cols: u32, log_n: u32, log_blowup: u32, trace_offset: u64, lde_offset: u64, d_out: *mut u32) -> c_int;
    pub fn smi_dev_hash_leaves(ctx: *mut smi_ctx, d_elems: *const u32, n: usize, d_digests: *mut u8) -> c_int;
    pub fn smi_dev_merkle_build(ctx: *mut smi_ctx, d_elems: *const u32, n: usize, d_nodes: *mut u8) -> c_int;
    pub fn smi_dev_merkle_build_rows(ctx: *mut smi_ctx, d_cols: *const u32, n_cols: u32, col_stride: usize, n: usize, d_nodes: *mut u8) -> c_int;
    pub fn smi_dev_merkle_from_digests(ctx: *mut smi_ctx, n: usize, d_nodes: *mut u8) -> c_int;
    pub fn smi_dev_hash_bytes(ctx: *mut smi_ctx, d_msg: *const u8, len: usize, d_out32: *mut u8) -> c_int;
    pub fn smi_dev_fri_fold(ctx: *mut smi_ctx, d_in: *const u32, len: usize, d_alpha: *const u64, offset: u64, omega: u64, d_out: *mut u32) -> c_int;
    pub fn smi_dev_fri_fold_shard(ctx: *mut smi_ctx, d_lo: *const u32, d_hi: *const u32, count: usize, index0: usize, full_len: usize, d_alpha: *const u64, offset: u64, omega: u64, d_out: *mut u32) -> c_int;
    pub fn smi_dev_fri_prove(ctx: *mut smi_ctx, cfg: *const smi_fri_cfg, d_codeword: *const u32, len: usize, proof: *mut *mut u8, proof_len: *mut usize, top_indices: *mut u64, run: *mut *mut smi_fri_run) -> c_int;
    pub fn smi_dev_combine_columns(ctx: *mut smi_ctx, d_cols: *const u32, n_cols: u32, len: usize, stride: usize, d_weights: *const u64, d_out: *mut u32) -> c_int;
    pub fn smi_dev_stark_prove(ctx: *mut smi_ctx, cfg: *const smi_stark_cfg, d_trace_cols: *const u32, column_roots: *mut u8, proof: *mut *mut u8, proof_len: *mut usize, top_indices: *mut u64, stage_ms: *mut f64) -> c_int;
    pub fn smi_stark_verify(ctx: *mut smi_ctx, cfg: *const smi_stark_cfg, column_roots: *const u8, proof: *const u8, proof_len: usize, accept: *mut c_int) -> c_int;
    pub fn smi_mgpu_unique_id(id: *mut u8) -> c_int;
    pub fn smi_mgpu_create(ctx: *mut smi_ctx, id: *const u8, rank: c_int, world: c_int, out: *mut *mut smi_mgpu) -> c_int;
    pub fn smi_mgpu_create_with(ctx: *mut smi_ctx, ops: *const smi_mgpu_coll, rank: c_int, world: c_int, out: *mut *mut smi_mgpu) -> c_int;
    pub fn smi_mgpu_destroy(m: *mut smi_mgpu);
    pub fn smi_mgpu_set_min_block(m: *mut smi_mgpu, min_block: usize) -> c_int;
    pub fn smi_mgpu_fri_commit(m: *mut smi_mgpu, cfg: *const smi_fri_cfg, d_block: *const u32, block_len: usize, roots: *mut u8, alphas: *mut u64, last_codeword: *mut u64, last_len: *mut usize) -> c_int;
    pub fn smi_mgpu_fri_prove(m: *mut smi_mgpu, cfg: *const smi_fri_cfg, d_block: *const u32, block_len: usize, proof: *mut *mut u8, proof_len: *mut usize, top_indices: *mut u64) -> c_int;
    pub fn smi_mgpu_lde(m: *mut smi_mgpu, d_trace_cols: *const u32, n_cols: u32, log_n: u32, log_blowup: u32, trace_offset: u64, lde_offset: u64, d_out_blocks: *mut u32) -> c_int;
    pub fn smi_mgpu_ntt(m: *mut smi_mgpu, d_strip: *mut u32, d_out: *mut u32, log_n: u32, inverse: c_int, offset: u64) -> c_int;
    pub fn smi_mgpu_ntt_natural(m: *mut smi_mgpu, d_strip: *mut u32, d_out: *mut u32, log_n: u32, inverse: c_int, offset: u64) -> c_int;
    pub fn smi_mgpu_ntt_first_digit(log_n: u32, log_r0: *mut u32) -> c_int;
    pub fn smi_mgpu_stark_prove(m: *mut smi_mgpu, cfg: *const smi_stark_cfg, d_trace_cols: *const u32, column_roots: *mut u8, proof: *mut *mut u8, proof_len: *mut usize, top_indices: *mut u64) -> c_int;
}
// END GENERATED

// ------------------------------------------------------------------------------------------------
// Safe layer.  One context per thread (the C context is single-owner: `!Sync`, src: SURVEY 8b
// "Threading"); (998244353, 3) is the reference field (src/ff.rs:191-197, 215-223).

pub const P_REF: u64 = 998_244_353;
pub const G_REF: u64 = 3;

/// Owning handle of an `smi_ctx` (`smi_ctx_destroy` on drop).  Not `Send`/`Sync`: raw pointer inside.
pub struct Context {
    raw: *mut smi_ctx,
}

impl Context {
    /// `smi_ctx_create(p, g, device)`; panics with the library's message when there is no GPU or the
    /// modulus is unsupported -- there is no CPU fallback behind this boundary.
    pub fn new(p: u64, g: u64, device: i32) -> Context {
        let mut raw: *mut smi_ctx = std::ptr::null_mut();
        check(unsafe { smi_ctx_create(p, g, device as c_int, &mut raw) });
        Context { raw }
    }
    pub fn reference_field() -> Context {
        Context::new(P_REF, G_REF, 0)
    }
    pub fn as_ptr(&self) -> *mut smi_ctx {
        self.raw
    }
    /// status -> panic with the reference's text, plus `smi_last_error` for the codes that carry detail
    pub fn check(&self, status: c_int) {
        if status == 0 {
            return;
        }
        let msg = status_text(status);
        if status <= SMI_ERR_BAD_ARG {
            let detail = unsafe { CStr::from_ptr(smi_last_error(self.raw)) }.to_string_lossy().into_owned();
            if !detail.is_empty() {
                panic!("{}: {}", msg, detail);
            }
        }
        panic!("{}", msg);
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { smi_ctx_destroy(self.raw) }
    }
}

thread_local! {
    static CTX: Context = Context::reference_field();
}

/// Runs `f` with this thread's context for the reference field.
pub fn with_ctx<R>(f: impl FnOnce(&Context) -> R) -> R {
    CTX.with(|c| f(c))
}

pub fn status_text(status: c_int) -> String {
    unsafe { CStr::from_ptr(smi_status_string(status)) }.to_string_lossy().into_owned()
}

/// A non-zero status becomes the reference's own panic: `smi_status_string` returns the identical message
/// ("no inverse", "Number of leaves must be power of 2", ...).
pub fn check(status: c_int) {
    if status != 0 {
        panic!("{}", status_text(status));
    }
}

fn log2_exact(n: usize) -> u32 {
    assert!(n.is_power_of_two(), "n must be a power of two");
    n.trailing_zeros()
}

/// `Some(offset)` when `domain` is `offset * omega^k` for the primitive `domain.len()`-th root the reference's
/// `prim_nth_root` returns -- the only domains the transforms serve; `None` means "keep the CPU body".
pub fn geometric_offset(ctx: &Context, domain: &[u64]) -> Option<u64> {
    let mut offset = 0u64;
    match unsafe { smi_domain_is_geometric(ctx.raw, domain.as_ptr(), domain.len(), &mut offset) } {
        0 => Some(offset),
        SMI_ERR_NOT_GEOMETRIC => None,
        st => {
            ctx.check(st);
            None
        }
    }
}

/// `Polynomial::interpolate_domain` on a geometric domain: coefficients (ascending) of the polynomial through
/// `(offset * omega^k, values[k])`.  The caller keeps the reference's two asserts and its H8 shape rule
/// (all-zero values of length > 1 give an empty coefficient vector).
pub fn interpolate_domain(ctx: &Context, values: &[u64], offset: u64) -> Vec<u64> {
    let mut coeffs = vec![0u64; values.len()];
    ctx.check(unsafe { smi_intt(ctx.raw, values.as_ptr(), coeffs.as_mut_ptr(), log2_exact(values.len()), offset) });
    coeffs
}

/// `Polynomial::eval_domain` on `offset * <omega_N>`, `N = 2^log_n >= coeffs.len()`, in domain order.
pub fn eval_domain(ctx: &Context, coeffs: &[u64], log_n: u32, offset: u64) -> Vec<u64> {
    let mut evals = vec![0u64; 1usize << log_n];
    ctx.check(unsafe { smi_coset_ntt(ctx.raw, coeffs.as_ptr(), coeffs.len(), evals.as_mut_ptr(), log_n, offset) });
    evals
}

/// `Polynomial::scale`: coefficient i times factor^i.
pub fn scale(ctx: &Context, coeffs: &[u64], factor: u64) -> Vec<u64> {
    let mut out = vec![0u64; coeffs.len()];
    ctx.check(unsafe { smi_poly_scale(ctx.raw, coeffs.as_ptr(), coeffs.len(), factor, out.as_mut_ptr()) });
    out
}

/// `Polynomial::mul` (empty for a zero operand, as the reference returns).
pub fn mul(ctx: &Context, a: &[u64], b: &[u64]) -> Vec<u64> {
    let mut out = vec![0u64; (a.len() + b.len()).max(1)];
    let mut n_out = 0usize;
    ctx.check(unsafe { smi_poly_mul(ctx.raw, a.as_ptr(), a.len(), b.as_ptr(), b.len(), out.as_mut_ptr(), &mut n_out) });
    out.truncate(n_out);
    out
}

/// `Polynomial::div` -> (quotient, remainder); a zero divisor panics "No division by zero".
pub fn div(ctx: &Context, a: &[u64], b: &[u64]) -> (Vec<u64>, Vec<u64>) {
    let mut q = vec![0u64; a.len().max(1)];
    let mut r = vec![0u64; a.len().max(b.len()).max(1)];
    let (mut nq, mut nr) = (0usize, 0usize);
    ctx.check(unsafe {
        smi_poly_div(ctx.raw, a.as_ptr(), a.len(), b.as_ptr(), b.len(), q.as_mut_ptr(), &mut nq, r.as_mut_ptr(), &mut nr)
    });
    q.truncate(nq);
    r.truncate(nr);
    (q, r)
}

/// `codeword.iter().map(|e| Hash::from_field_elements(&[e.value]))` in one call (src/fri.rs:118-121).
pub fn hash_leaves(ctx: &Context, elems: &[u64]) -> Vec<[u8; 32]> {
    let mut out = vec![[0u8; 32]; elems.len()];
    ctx.check(unsafe { smi_hash_leaves(ctx.raw, elems.as_ptr(), elems.len(), out.as_mut_ptr() as *mut u8) });
    out
}

/// `Hash::combine` over adjacent pairs: out[i] = combine(digests[2i], digests[2i+1]).
pub fn combine_pairs(ctx: &Context, digests: &[[u8; 32]]) -> Vec<[u8; 32]> {
    let n_pairs = digests.len() / 2;
    let mut out = vec![[0u8; 32]; n_pairs];
    ctx.check(unsafe { smi_hash_combine_pairs(ctx.raw, digests.as_ptr() as *const u8, n_pairs, out.as_mut_ptr() as *mut u8) });
    out
}

/// `Hash::from_bytes` of one message.
pub fn hash_bytes(ctx: &Context, msg: &[u8]) -> [u8; 32] {
    let mut out = [0u8; 32];
    ctx.check(unsafe { smi_hash_bytes(ctx.raw, msg.as_ptr(), msg.len(), out.as_mut_ptr()) });
    out
}

/// `MerkleTree` with all levels resident on the device; `open` is a gather, nothing is rebuilt.
pub struct MerkleTree<'c> {
    ctx: &'c Context,
    raw: *mut smi_tree,
}

impl<'c> MerkleTree<'c> {
    /// `MerkleTree::new(&leaves)`: panics "Cannot create tree from empty leaves" / "Number of leaves must be power of 2".
    pub fn new(ctx: &'c Context, leaves: &[[u8; 32]]) -> MerkleTree<'c> {
        let mut raw: *mut smi_tree = std::ptr::null_mut();
        ctx.check(unsafe { smi_merkle_new(ctx.raw, leaves.as_ptr() as *const u8, leaves.len(), &mut raw) });
        MerkleTree { ctx, raw }
    }
    /// the same tree from the codeword itself (leaf hashing fused with the bottom levels)
    pub fn from_codeword(ctx: &'c Context, codeword: &[u64]) -> MerkleTree<'c> {
        let mut raw: *mut smi_tree = std::ptr::null_mut();
        ctx.check(unsafe { smi_merkle_from_codeword(ctx.raw, codeword.as_ptr(), codeword.len(), &mut raw) });
        MerkleTree { ctx, raw }
    }
    pub fn get_root(&self) -> [u8; 32] {
        let mut root = [0u8; 32];
        self.ctx.check(unsafe { smi_merkle_root(self.ctx.raw, self.raw, root.as_mut_ptr()) });
        root
    }
    /// `open(index)`: the sibling at each level, bottom up; panics "Index out of bounds".
    pub fn open(&self, index: usize) -> Vec<[u8; 32]> {
        let n = unsafe { smi_merkle_num_leaves(self.raw) };
        let mut path = vec![[0u8; 32]; (usize::BITS - n.leading_zeros()) as usize];
        let mut depth = 0usize;
        self.ctx.check(unsafe { smi_merkle_open(self.ctx.raw, self.raw, index, path.as_mut_ptr() as *mut u8, &mut depth) });
        path.truncate(depth);
        path
    }
    /// `nodes[level]` of the reference's struct, on demand
    pub fn level(&self, level: u32) -> Vec<[u8; 32]> {
        let n = unsafe { smi_merkle_num_leaves(self.raw) } >> level;
        let mut out = vec![[0u8; 32]; n.max(1)];
        let mut n_out = 0usize;
        self.ctx.check(unsafe { smi_merkle_level(self.ctx.raw, self.raw, level, out.as_mut_ptr() as *mut u8, &mut n_out) });
        out.truncate(n_out);
        out
    }
    /// `MerkleTree::commit(&leaves)` without keeping the tree
    pub fn commit(ctx: &Context, leaves: &[[u8; 32]]) -> [u8; 32] {
        let mut root = [0u8; 32];
        ctx.check(unsafe { smi_merkle_commit(ctx.raw, leaves.as_ptr() as *const u8, leaves.len(), root.as_mut_ptr()) });
        root
    }
}

impl Drop for MerkleTree<'_> {
    fn drop(&mut self) {
        unsafe { smi_merkle_free(self.raw) }
    }
}

/// `Fri::new` checks (src/fri.rs:37-45) + the struct the entry points take.
pub fn fri_cfg(ctx: &Context, omega: u64, offset: u64, domain_length: u64, expansion_factor: u64, num_colinearity_tests: u64) -> smi_fri_cfg {
    let cfg = smi_fri_cfg { omega, offset, domain_length, expansion_factor, num_colinearity_tests };
    ctx.check(unsafe { smi_fri_check(ctx.raw, &cfg) });
    cfg
}

/// `Fri::fold_codeword(codeword, alpha, offset, omega)`; alpha is passed unreduced, as `FiatShamir::challenge` makes it.
pub fn fold_codeword(ctx: &Context, codeword: &[u64], alpha: u64, offset: u64, omega: u64) -> Vec<u64> {
    let mut out = vec![0u64; codeword.len() / 2];
    ctx.check(unsafe { smi_fri_fold(ctx.raw, codeword.as_ptr(), codeword.len(), alpha, offset, omega, out.as_mut_ptr()) });
    out
}

/// `Fri::prove` with a fresh `FiatShamir` and `ProofStream`: (`ProofStream::serialize()` bytes, top-level indices).
/// The caller deserializes, pushes the objects and absorbs the roots to leave its own objects as the reference does.
pub fn fri_prove(ctx: &Context, cfg: &smi_fri_cfg, codeword: &[u64]) -> (Vec<u8>, Vec<usize>) {
    let mut proof: *mut u8 = std::ptr::null_mut();
    let mut len = 0usize;
    let mut top = vec![0u64; (cfg.num_colinearity_tests as usize).max(1)];
    ctx.check(unsafe { smi_fri_prove(ctx.raw, cfg, codeword.as_ptr(), codeword.len(), &mut proof, &mut len, top.as_mut_ptr()) });
    let bytes = unsafe { std::slice::from_raw_parts(proof, len) }.to_vec();
    unsafe { smi_free(proof as *mut c_void) };
    top.truncate(cfg.num_colinearity_tests as usize);
    (bytes, top.into_iter().map(|v| v as usize).collect())
}

/// `Fri::verify` of a serialized stream against a fresh transcript: (verdict, the (index, value) pairs the
/// reference pushes to `polynomial_values`).  `Err(())` = the last layer's domain is not a coset of roots of
/// unity (SMI_ERR_NOT_GEOMETRIC): run the CPU body instead.  A reference panic panics here with the same text.
pub fn fri_verify(ctx: &Context, cfg: &smi_fri_cfg, proof: &[u8]) -> Result<(bool, Vec<(usize, u64)>), ()> {
    let t = cfg.num_colinearity_tests as usize;
    let (mut idx, mut val) = (vec![0u64; 2 * t + 2], vec![0u64; 2 * t + 2]);
    let (mut accept, mut n) = (0 as c_int, 0usize);
    let st = unsafe { smi_fri_verify(ctx.raw, cfg, proof.as_ptr(), proof.len(), &mut accept, idx.as_mut_ptr(), val.as_mut_ptr(), &mut n) };
    if st == SMI_ERR_NOT_GEOMETRIC {
        return Err(());
    }
    ctx.check(st);
    Ok((accept != 0, (0..n).map(|i| (idx[i] as usize, val[i])).collect()))
}

/// `Trace::to_field_elements` for all columns at once: row-major i128 rows (`Vec<Vec<i128>>` flattened) to
/// column-major u64 residues.
pub fn trace_pack(ctx: &Context, rows: &[i128], n_rows: usize, n_cols: usize) -> Vec<u64> {
    assert!(rows.len() == n_rows * n_cols);
    let mut cols = vec![0u64; n_rows * n_cols];
    ctx.check(unsafe { smi_trace_pack(ctx.raw, rows.as_ptr() as *const c_void, n_rows, n_cols, cols.as_mut_ptr()) });
    cols
}

/// Low-degree extension of `n_cols` columns (column-major, 2^log_n rows each): interpolate on
/// `trace_offset * <omega_n>`, evaluate on `lde_offset * <omega_N>`, N = n << log_blowup.
pub fn lde(ctx: &Context, cols: &[u64], n_cols: u32, log_n: u32, log_blowup: u32, trace_offset: u64, lde_offset: u64) -> Vec<u64> {
    assert!(cols.len() == (n_cols as usize) << log_n);
    let mut out = vec![0u64; (n_cols as usize) << (log_n + log_blowup)];
    ctx.check(unsafe { smi_lde(ctx.raw, cols.as_ptr(), n_cols, log_n, log_blowup, trace_offset, lde_offset, out.as_mut_ptr()) });
    out
}
