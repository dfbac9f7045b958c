/*
 * stark_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See stark_oracle.h for the contract and the parity-pin status.
 *
 * Restates /root/reference/src/{ff,utils,hash,merkle,fiat_shamir,stream,trace,fri}.rs
 * and src/univariate/{mod,add,sub,mul,div,exp,eval,interpolate}.rs op for op: the same
 * u128 `% p` arithmetic, the same recursive xgcd, the same Lagrange / power-sum /
 * per-element exp+xgcd fold / level-by-level Merkle algorithms, single-threaded.
 * Nothing here is copied text: the reference is Rust, this is plain C.
 */
#include "stark_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef __int128 i128;

/* ------------------------------------------------------------------ panics */
static char g_panic[256];
static char g_reject[256];

static void so_panic(const char *msg) {
    if (g_panic[0] == 0) { /* keep the first panic, like an unwinding thread would */
        strncpy(g_panic, msg, sizeof(g_panic) - 1);
        g_panic[sizeof(g_panic) - 1] = 0;
    }
}
const char *so_last_panic(void) { return g_panic; }
void so_clear_panic(void) { g_panic[0] = 0; g_reject[0] = 0; }
const char *so_fri_last_reject(void) { return g_reject; }
static void so_reject(const char *msg) {
    strncpy(g_reject, msg, sizeof(g_reject) - 1);
    g_reject[sizeof(g_reject) - 1] = 0;
}
void so_free(void *p) { free(p); }

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "stark_oracle: out of memory\n"); abort(); }
    return p;
}

/* ------------------------------------------------------------------- field */
/* ff.rs:146-152 */
uint64_t so_ff_add(uint64_t p, uint64_t l, uint64_t r) {
    return (uint64_t)(((u128)l + (u128)r) % (u128)p);
}
/* ff.rs:154-160 -- `p + l - r` in u128; wraps like a release build if r > p + l (H6) */
uint64_t so_ff_sub(uint64_t p, uint64_t l, uint64_t r) {
    return (uint64_t)((((u128)p + (u128)l) - (u128)r) % (u128)p);
}
/* ff.rs:138-144 */
uint64_t so_ff_mul(uint64_t p, uint64_t l, uint64_t r) {
    return (uint64_t)(((u128)l * (u128)r) % (u128)p);
}
/* ff.rs:162-167 */
uint64_t so_ff_neg(uint64_t p, uint64_t x) { return (uint64_t)(p - x) % p; }

/* utils.rs:3-13, recursive extended Euclid in i128 */
static void xgcd_rec(uint64_t x, uint64_t y, i128 *g, i128 *a, i128 *b) {
    if (y == 0) { *g = (i128)x; *a = 1; *b = 0; return; }
    i128 g1, x1, y1;
    xgcd_rec(y, x % y, &g1, &x1, &y1);
    *g = g1;
    *a = y1;
    *b = x1 - ((i128)x / (i128)y) * y1;
}
void so_xgcd(uint64_t x, uint64_t y, int64_t out[3]) {
    i128 g, a, b;
    xgcd_rec(x, y, &g, &a, &b);
    out[0] = (int64_t)g; out[1] = (int64_t)a; out[2] = (int64_t)b;
}
/* ff.rs:169-178 */
uint64_t so_ff_inv(uint64_t p, uint64_t x) {
    i128 g, a, b;
    xgcd_rec(x, p, &g, &a, &b);
    if (g != 1) { so_panic("no inverse"); return 0; }
    i128 pp = (i128)p;
    i128 inv = ((a % pp) + pp) % pp;
    return (uint64_t)inv;
}
/* ff.rs:181-189 */
uint64_t so_ff_div(uint64_t p, uint64_t l, uint64_t r) {
    if (r == 0) { so_panic("no division by zero"); return 0; }
    uint64_t rinv = so_ff_inv(p, r);
    return (uint64_t)(((u128)l * (u128)rinv) % (u128)p);
}
/* ff.rs:200-213 */
uint64_t so_ff_exp(uint64_t p, uint64_t base, uint64_t e) {
    uint64_t res = 1;
    while (e > 0) {
        if (e % 2 == 1) res = so_ff_mul(p, res, base);
        base = so_ff_mul(p, base, base);
        e >>= 1;
    }
    return res;
}
/* ff.rs:191-197 */
uint64_t so_ff_g(uint64_t p) {
    if (p != SO_P_REF) { so_panic("assertion failed: self.p == 998244353"); return 0; }
    return 3;
}
/* ff.rs:215-223 */
uint64_t so_ff_prim_nth_root(uint64_t p, uint64_t n) {
    if (p != SO_P_REF) { so_panic("assertion failed: self.p == 998244353"); return 0; }
    if (n == 0 || (n & (n - 1)) != 0) { so_panic("n must be a power of two"); return 0; }
    if (n > (1ULL << 23)) { so_panic("n > 2^23 not supported by this modulus"); return 0; }
    uint64_t g = so_ff_g(p);
    return so_ff_exp(p, g, (p - 1) / n);
}
uint64_t so_ff_prim_nth_root_g(uint64_t p, uint64_t g, uint64_t n) {
    if (n == 0 || (n & (n - 1)) != 0) { so_panic("n must be a power of two"); return 0; }
    if ((p - 1) % n != 0) { so_panic("n does not divide p-1"); return 0; }
    return so_ff_exp(p, g, (p - 1) / n);
}
/* ff.rs:225-232 */
uint64_t so_ff_sample(uint64_t p, const uint8_t *salt, size_t len) {
    uint64_t acc = 0;
    for (size_t i = 0; i < len; i++) {
        acc = (uint64_t)((((u128)acc) << 8) % (u128)p);
        acc = (uint64_t)((((u128)acc) ^ (u128)salt[i]) % (u128)p);
    }
    return acc;
}

/* ------------------------------------------------------------- polynomials */
/* mod.rs:54-68: two scans, exactly like the reference (all-zero test, then max idx) */
int64_t so_poly_deg(const uint64_t *c, size_t n) {
    if (n == 0) return -1;
    int all_zero = 1;
    for (size_t i = 0; i < n; i++) if (c[i] != 0) { all_zero = 0; break; }
    if (all_zero) return -1;
    size_t maxidx = 0;
    for (size_t i = 0; i < n; i++) if (c[i] != 0) maxidx = i;
    return (int64_t)maxidx;
}
/* mod.rs:13-39 */
int so_poly_eq(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb) {
    (void)p;
    int64_t da = so_poly_deg(a, na), db = so_poly_deg(b, nb);
    if (da != db) return 0;
    if (da == -1) return 1;
    for (int64_t i = 0; i <= da; i++) {
        uint64_t x = (size_t)i < na ? a[i] : 0, y = (size_t)i < nb ? b[i] : 0;
        if (x != y) return 0;
    }
    return 1;
}
/* add.rs:6-32 */
size_t so_poly_add(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out) {
    if (so_poly_deg(a, na) == -1) { memmove(out, b, nb * 8); return nb; }
    if (so_poly_deg(b, nb) == -1) { memmove(out, a, na * 8); return na; }
    size_t n = na > nb ? na : nb;
    for (size_t i = 0; i < n; i++) {
        uint64_t l = i < na ? a[i] : 0, r = i < nb ? b[i] : 0;
        out[i] = so_ff_add(p, l, r);
    }
    return n;
}
/* sub.rs:8-34 (lhs zero -> neg(rhs), mod.rs:70-75) */
size_t so_poly_sub(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out) {
    if (so_poly_deg(a, na) == -1) {
        for (size_t i = 0; i < nb; i++) out[i] = so_ff_neg(p, b[i]);
        return nb;
    }
    if (so_poly_deg(b, nb) == -1) { memmove(out, a, na * 8); return na; }
    size_t n = na > nb ? na : nb;
    for (size_t i = 0; i < n; i++) {
        uint64_t l = i < na ? a[i] : 0, r = i < nb ? b[i] : 0;
        out[i] = so_ff_sub(p, l, r);
    }
    return n;
}
/* mul.rs:6-29; out must not alias a or b */
size_t so_poly_mul(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *out) {
    if (so_poly_deg(a, na) == -1 || so_poly_deg(b, nb) == -1) return 0;
    size_t n = na + nb - 1;
    for (size_t i = 0; i < n; i++) out[i] = 0;
    for (size_t i = 0; i < na; i++) {
        if (a[i] == 0) continue;
        for (size_t j = 0; j < nb; j++)
            out[i + j] = so_ff_add(p, out[i + j], so_ff_mul(p, a[i], b[j]));
    }
    return n;
}
/* mod.rs:99-113 */
size_t so_poly_scale(uint64_t p, const uint64_t *a, size_t na, uint64_t factor, uint64_t *out) {
    for (size_t i = 0; i < na; i++) {
        uint64_t power = so_ff_exp(p, factor, (uint64_t)i);
        out[i] = so_ff_mul(p, power, a[i]);
    }
    return na;
}
/* eval.rs:6-14 */
uint64_t so_poly_eval(uint64_t p, const uint64_t *c, size_t n, uint64_t x) {
    uint64_t xi = 1, val = 0;
    for (size_t i = 0; i < n; i++) {
        val = so_ff_add(p, val, so_ff_mul(p, c[i], xi));
        xi = so_ff_mul(p, xi, x);
    }
    return val;
}
/* eval.rs:16-21 */
void so_poly_eval_domain(uint64_t p, const uint64_t *c, size_t n, const uint64_t *dom, size_t nd, uint64_t *out) {
    for (size_t k = 0; k < nd; k++) out[k] = so_poly_eval(p, c, n, dom[k]);
}
/* interpolate.rs:6-44 */
size_t so_poly_interpolate_domain(uint64_t p, const uint64_t *dom, const uint64_t *vals, size_t n, uint64_t *out) {
    if (n == 0) { so_panic("assertion failed: domain.len() > 0"); return 0; }
    uint64_t x[2] = {0, 1};
    uint64_t *acc = xmalloc((n + 2) * 8), *prod = xmalloc((n + 2) * 8), *tmp = xmalloc((n + 2) * 8);
    size_t nacc = 1, nprod, ntmp;
    acc[0] = 0;
    for (size_t i = 0; i < n; i++) {
        prod[0] = vals[i]; nprod = 1;
        for (size_t j = 0; j < n; j++) {
            if (j == i) continue;
            uint64_t xj[1] = {dom[j]};
            uint64_t denom = so_ff_inv(p, so_ff_sub(p, dom[i], dom[j]));   /* :34 */
            if (g_panic[0]) { free(acc); free(prod); free(tmp); return 0; }
            uint64_t lin[2];
            size_t nlin = so_poly_sub(p, x, 2, xj, 1, lin);                /* &x - &xj */
            ntmp = so_poly_mul(p, prod, nprod, lin, nlin, tmp);            /* :35 */
            uint64_t *sw = prod; prod = tmp; tmp = sw; nprod = ntmp;
            for (size_t c = 0; c < nprod; c++) prod[c] = so_ff_mul(p, prod[c], denom); /* :37-39 */
        }
        ntmp = so_poly_add(p, acc, nacc, prod, nprod, tmp);                /* :41 */
        uint64_t *sw = acc; acc = tmp; tmp = sw; nacc = ntmp;
    }
    memcpy(out, acc, nacc * 8);
    free(acc); free(prod); free(tmp);
    return nacc;
}
/* mod.rs:77-96 */
size_t so_poly_zerofier(uint64_t p, const uint64_t *dom, size_t n, uint64_t *out) {
    uint64_t x[2] = {0, 1};
    uint64_t *acc = xmalloc((n + 2) * 8), *tmp = xmalloc((n + 2) * 8);
    size_t nacc = 1; acc[0] = 1;
    for (size_t k = 0; k < n; k++) {
        uint64_t d[1] = {dom[k]}, lin[2];
        size_t nlin = so_poly_sub(p, x, 2, d, 1, lin);
        size_t nt = so_poly_mul(p, acc, nacc, lin, nlin, tmp);
        uint64_t *sw = acc; acc = tmp; tmp = sw; nacc = nt;
    }
    memcpy(out, acc, nacc * 8);
    free(acc); free(tmp);
    return nacc;
}
/* div.rs:6-41 */
void so_poly_div(uint64_t p, const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                 uint64_t *q_out, size_t *nq, uint64_t *r_out, size_t *nr) {
    int64_t db = so_poly_deg(b, nb), da = so_poly_deg(a, na);
    if (db == -1) { so_panic("No division by zero"); *nq = 0; *nr = 0; return; }
    if (da < db) { *nq = 0; memmove(r_out, a, na * 8); *nr = na; return; }
    size_t qn = (size_t)(da - db + 1);
    for (size_t i = 0; i < qn; i++) q_out[i] = 0;
    size_t cap = na + nb + 2;
    uint64_t *r = xmalloc(cap * 8), *sub = xmalloc(cap * 8), *shift = xmalloc(cap * 8), *tmp = xmalloc(cap * 8);
    size_t rn = na; memcpy(r, a, na * 8);
    int64_t dr;
    while ((dr = so_poly_deg(r, rn)) >= db) {
        uint64_t coeff = so_ff_div(p, r[dr], b[db]);
        size_t sh = (size_t)(dr - db);
        for (size_t i = 0; i < sh; i++) shift[i] = 0;
        shift[sh] = coeff;
        size_t sn = so_poly_mul(p, shift, sh + 1, b, nb, sub);
        q_out[sh] = coeff;
        size_t tn = so_poly_sub(p, r, rn, sub, sn, tmp);
        uint64_t *sw = r; r = tmp; tmp = sw; rn = tn;
    }
    *nq = qn;
    memcpy(r_out, r, rn * 8); *nr = rn;
    free(r); free(sub); free(shift); free(tmp);
}
/* exp.rs:6-33 */
size_t so_poly_exp(uint64_t p, const uint64_t *a, size_t na, uint64_t e, uint64_t *out, size_t out_cap) {
    if (e == 0) { out[0] = 1; return 1; }
    if (so_poly_deg(a, na) == -1) return 0;
    uint64_t *res = xmalloc(out_cap * 8), *bp = xmalloc(out_cap * 8), *tmp = xmalloc(out_cap * 8);
    size_t rn = 1, bn = na; res[0] = 1; memcpy(bp, a, na * 8);
    while (e != 0) {
        if (e & 1) {
            size_t tn = so_poly_mul(p, res, rn, bp, bn, tmp);
            uint64_t *sw = res; res = tmp; tmp = sw; rn = tn;
        }
        e >>= 1;
        /* the reference squares once more after the last bit; the value is unused, and
         * the oracle skips it only when the square would outgrow the caller's buffer */
        if (2 * bn - 1 <= out_cap) {
            size_t tn = so_poly_mul(p, bp, bn, bp, bn, tmp);
            uint64_t *sw = bp; bp = tmp; tmp = sw; bn = tn;
        } else if (e != 0) { so_panic("so_poly_exp: out_cap too small"); break; }
    }
    memcpy(out, res, rn * 8);
    free(res); free(bp); free(tmp);
    return rn;
}
/* mod.rs:145-152 */
int so_poly_test_colinearity(uint64_t p, const uint64_t *xs, const uint64_t *ys, size_t n) {
    if (n < 2) { so_panic("At least 2 points to test colinearity"); return 0; }
    uint64_t *c = xmalloc((n + 2) * 8);
    size_t nc = so_poly_interpolate_domain(p, xs, ys, n, c);
    int ok = so_poly_deg(c, nc) <= 1;
    free(c);
    return ok;
}

/* -------------------------------------------------------------------- hash */
static const uint8_t PRIMES[16] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53}; /* hash.rs:53 */
static const uint8_t ROUND_CONSTANTS[32] = {                                                    /* hash.rs:96-99 */
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36, 0x6c, 0xd8, 0xab, 0x4d, 0x9a, 0x2f,
    0x5e, 0xbc, 0x63, 0xc6, 0x97, 0x35, 0x6a, 0xd4, 0xb3, 0x7d, 0xfa, 0xef, 0xc5, 0x91, 0x39, 0x72};

static inline uint8_t rotl8(uint8_t b, unsigned n) { return (uint8_t)((b << n) | (b >> (8 - n))); } /* hash.rs:55-57 */
static inline uint8_t sbox(uint8_t b) {                                                              /* hash.rs:88-94 */
    uint8_t r = (uint8_t)(b * 251u);
    r = rotl8(r, 1);
    r ^= 0x63;
    return r;
}
/* hash.rs:59-86 */
static void mix_state(uint8_t s[32]) {
    for (int i = 0; i < 32; i++) s[i] = sbox(s[i]);
    for (int i = 0; i < 8; i++) {
        int b = i * 4;
        uint8_t t0 = s[b], t1 = s[b + 1], t2 = s[b + 2], t3 = s[b + 3];
        s[b] = t0 ^ t1 ^ t3;
        s[b + 1] = t0 ^ t2 ^ t3;
        s[b + 2] = t0 ^ t1 ^ t2;
        s[b + 3] = t1 ^ t2 ^ t3;
    }
    for (int i = 0; i < 32; i++) { /* in place, ascending: prev already updated for i>=1 */
        int next = (i + 1) % 32;
        int prev = i == 0 ? 31 : i - 1;
        s[i] = (uint8_t)(s[i] + s[next] + s[prev]);
    }
    for (int i = 0; i < 32; i++) s[i] = (uint8_t)(s[i] + ROUND_CONSTANTS[i]);
}
/* hash.rs:7-30 */
void so_hash_from_bytes(const uint8_t *bytes, size_t len, uint8_t out[32]) {
    uint8_t s[32];
    for (int i = 0; i < 32; i++) s[i] = PRIMES[i % 16];
    size_t nchunks = (len + 31) / 32; /* bytes.chunks(32): empty input -> zero chunks */
    for (size_t ci = 0; ci < nchunks; ci++) {
        size_t off = ci * 32, clen = len - off < 32 ? len - off : 32;
        for (size_t i = 0; i < clen; i++) {
            size_t pos = (i + ci * 32) % 32;
            s[pos] = (uint8_t)(s[pos] + bytes[off + i]);
            s[pos] = rotl8(s[pos], 3);
            s[(pos + 7) % 32] ^= s[pos];
        }
        mix_state(s);
    }
    for (int k = 0; k < 8; k++) mix_state(s);
    memcpy(out, s, 32);
}
/* hash.rs:32-35 */
void so_hash_from_field_elements(const uint64_t *e, size_t n, uint8_t out[32]) {
    uint8_t *buf = xmalloc(n * 8);
    for (size_t i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) buf[i * 8 + b] = (uint8_t)(e[i] >> (8 * b)); /* to_le_bytes */
    so_hash_from_bytes(buf, n * 8, out);
    free(buf);
}
/* hash.rs:37-39 */
void so_hash_from_u64(uint64_t v, uint8_t out[32]) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    so_hash_from_bytes(b, 8, out);
}
/* hash.rs:41-46 */
void so_hash_combine(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]) {
    uint8_t c[64];
    memcpy(c, l, 32); memcpy(c + 32, r, 32);
    so_hash_from_bytes(c, 64, out);
}

/* ------------------------------------------------------------------ merkle */
static int is_pow2(size_t n) { return n != 0 && (n & (n - 1)) == 0; }

/* merkle.rs:11-38 */
int so_merkle_new(const uint8_t *leaves, size_t n, uint8_t *nodes_out) {
    if (n == 0) { so_panic("Cannot create tree from empty leaves"); return -1; }
    if (!is_pow2(n)) { so_panic("Number of leaves must be power of 2"); return -1; }
    memcpy(nodes_out, leaves, n * 32);
    uint8_t *cur = nodes_out;
    size_t len = n;
    while (len > 1) {
        uint8_t *next = cur + len * 32;
        for (size_t i = 0; i < len; i += 2) so_hash_combine(cur + i * 32, cur + (i + 1) * 32, next + (i / 2) * 32);
        cur = next; len /= 2;
    }
    return 0;
}
/* merkle.rs:44-65 */
int so_merkle_commit(const uint8_t *leaves, size_t n, uint8_t root[32]) {
    if (n == 0) { so_panic("Cannot create tree from empty leaves"); return -1; }
    if (!is_pow2(n)) { so_panic("Number of leaves must be power of 2"); return -1; }
    uint8_t *nodes = xmalloc((2 * n - 1) * 32);
    so_merkle_new(leaves, n, nodes);
    memcpy(root, nodes + (2 * n - 2) * 32, 32);
    free(nodes);
    return 0;
}
/* merkle.rs:67-80 */
int so_merkle_open(const uint8_t *nodes, size_t n, size_t index, uint8_t *path_out) {
    if (index >= n) { so_panic("Index out of bounds"); return -1; }
    size_t idx = index, len = n, depth = 0;
    const uint8_t *lvl = nodes;
    while (len > 1) { /* for level in 0..nodes.len()-1 */
        size_t sib = (idx % 2 == 0) ? idx + 1 : idx - 1;
        memcpy(path_out + depth * 32, lvl + sib * 32, 32);
        idx /= 2; lvl += len * 32; len /= 2; depth++;
    }
    return (int)depth;
}
/* merkle.rs:82-96 */
int so_merkle_verify(const uint8_t leaf[32], size_t index, const uint8_t *path, size_t depth, const uint8_t root[32]) {
    uint8_t cur[32], nxt[32];
    memcpy(cur, leaf, 32);
    size_t idx = index;
    for (size_t d = 0; d < depth; d++) {
        if (idx % 2 == 0) so_hash_combine(cur, path + d * 32, nxt);
        else so_hash_combine(path + d * 32, cur, nxt);
        memcpy(cur, nxt, 32);
        idx /= 2;
    }
    return memcmp(cur, root, 32) == 0;
}

/* ------------------------------------------------------------- fiat-shamir */
struct so_fs { uint8_t *t; size_t len, cap; };
so_fs *so_fs_new(void) { so_fs *f = xmalloc(sizeof *f); f->t = NULL; f->len = f->cap = 0; return f; }
void so_fs_free(so_fs *f) { if (f) { free(f->t); free(f); } }
/* fiat_shamir.rs:15-17 */
void so_fs_absorb(so_fs *f, const uint8_t *data, size_t len) {
    if (f->len + len > f->cap) {
        f->cap = (f->len + len) * 2 + 64;
        f->t = realloc(f->t, f->cap);
        if (!f->t) abort();
    }
    memcpy(f->t + f->len, data, len);
    f->len += len;
}
/* fiat_shamir.rs:19-25: hash of the WHOLE transcript, first 8 bytes LE, unreduced */
uint64_t so_fs_challenge(const so_fs *f) {
    uint8_t h[32];
    so_hash_from_bytes(f->t, f->len, h);
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)h[i] << (8 * i);
    return v;
}

/* ------------------------------------------------------------------- trace */
void so_trace_to_field_elements(const uint64_t *lo, const uint64_t *hi, size_t n, uint64_t *out) {
    (void)hi; /* trace.rs:32 `e as u64` keeps the low 64 bits, no reduction */
    for (size_t i = 0; i < n; i++) out[i] = lo[i];
}
/* trace.rs:36-49 */
void so_trace_fibonacci(size_t length, uint64_t *out_lo, uint64_t *out_hi) {
    i128 a = 1, b = 1;
    for (size_t i = 0; i < length; i++) {
        out_lo[i] = (uint64_t)(u128)a;
        out_hi[i] = (uint64_t)((u128)a >> 64);
        i128 next = (i128)((u128)a + (u128)b); /* i128 overflow past row ~182 panics in debug */
        a = b; b = next;
    }
}

/* ------------------------------------------------------------ proof stream */
/* stream.rs:8-14 */
enum { PO_ROOT = 0, PO_FE = 1, PO_FES = 2, PO_PATH = 3 };
typedef struct {
    int tag;
    uint8_t hash[32];
    uint64_t fe;
    uint64_t *fes; size_t nfes;
    uint8_t *path; size_t npath;
} pobj;
typedef struct { pobj *o; size_t n, cap, head; } pstream;

static void ps_init(pstream *s) { s->o = NULL; s->n = s->cap = s->head = 0; }
static void ps_free(pstream *s) {
    for (size_t i = 0; i < s->n; i++) { free(s->o[i].fes); free(s->o[i].path); }
    free(s->o);
}
static pobj *ps_slot(pstream *s) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->o = realloc(s->o, s->cap * sizeof(pobj)); if (!s->o) abort(); }
    pobj *o = &s->o[s->n++];
    memset(o, 0, sizeof *o);
    return o;
}
static void ps_push_root(pstream *s, const uint8_t h[32]) { pobj *o = ps_slot(s); o->tag = PO_ROOT; memcpy(o->hash, h, 32); }
static void ps_push_fes(pstream *s, const uint64_t *v, size_t n) {
    pobj *o = ps_slot(s); o->tag = PO_FES; o->nfes = n; o->fes = xmalloc(n * 8); memcpy(o->fes, v, n * 8);
}
static void ps_push_path(pstream *s, const uint8_t *p, size_t n) {
    pobj *o = ps_slot(s); o->tag = PO_PATH; o->npath = n; o->path = xmalloc(n * 32); memcpy(o->path, p, n * 32);
}
/* stream.rs:27-33 pop = remove(0) */
static pobj *ps_pop(pstream *s) { return s->head < s->n ? &s->o[s->head++] : NULL; }

static void put_u64(uint8_t **w, uint64_t v) { for (int i = 0; i < 8; i++) *(*w)++ = (uint8_t)(v >> (8 * i)); }
static uint64_t get_u64(const uint8_t *b) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)b[i] << (8 * i); return v; }

/* stream.rs:35-64 */
static uint8_t *ps_serialize(const pstream *s, size_t *len_out) {
    size_t len = 0;
    for (size_t i = 0; i < s->n; i++) {
        const pobj *o = &s->o[i];
        switch (o->tag) {
        case PO_ROOT: len += 1 + 32; break;
        case PO_FE: len += 1 + 8; break;
        case PO_FES: len += 1 + 8 + 8 * o->nfes; break;
        case PO_PATH: len += 1 + 8 + 32 * o->npath; break;
        }
    }
    uint8_t *buf = xmalloc(len), *w = buf;
    for (size_t i = 0; i < s->n; i++) {
        const pobj *o = &s->o[i];
        *w++ = (uint8_t)o->tag;
        switch (o->tag) {
        case PO_ROOT: memcpy(w, o->hash, 32); w += 32; break;
        case PO_FE: put_u64(&w, o->fe); break;
        case PO_FES: put_u64(&w, o->nfes); for (size_t k = 0; k < o->nfes; k++) put_u64(&w, o->fes[k]); break;
        case PO_PATH: put_u64(&w, o->npath); memcpy(w, o->path, 32 * o->npath); w += 32 * o->npath; break;
        }
    }
    *len_out = len;
    return buf;
}
/* stream.rs:66-168, including its leniency on truncated input */
static void ps_deserialize(pstream *s, const uint8_t *b, size_t n) {
    ps_init(s);
    size_t i = 0;
    while (i < n) {
        uint8_t tag = b[i]; i += 1;
        if (tag == 0) {
            if (i + 32 <= n) { ps_push_root(s, b + i); i += 32; }
        } else if (tag == 1) {
            if (i + 8 <= n) { pobj *o = ps_slot(s); o->tag = PO_FE; o->fe = get_u64(b + i); i += 8; }
        } else if (tag == 2) {
            if (i + 8 <= n) {
                uint64_t len = get_u64(b + i); i += 8;
                size_t avail = (n - i) / 8, take = len < avail ? (size_t)len : avail; /* loop does nothing once short */
                pobj *o = ps_slot(s); o->tag = PO_FES; o->nfes = take; o->fes = xmalloc(take * 8);
                for (size_t k = 0; k < take; k++) { o->fes[k] = get_u64(b + i); i += 8; }
            }
        } else if (tag == 3) {
            if (i + 8 <= n) {
                uint64_t len = get_u64(b + i); i += 8;
                size_t avail = (n - i) / 32, take = len < avail ? (size_t)len : avail;
                pobj *o = ps_slot(s); o->tag = PO_PATH; o->npath = take; o->path = xmalloc(take * 32);
                memcpy(o->path, b + i, take * 32); i += take * 32;
            }
        } else break;
    }
}

/* --------------------------------------------------------------------- FRI */
/* fri.rs:37-45 */
int so_fri_new_check(const so_fri_cfg *c) {
    if (!is_pow2(c->domain_length)) { so_panic("Domain length must be power of 2"); return -1; }
    if (!is_pow2(c->expansion_factor)) { so_panic("Expansion factor must be power of 2"); return -1; }
    if (c->expansion_factor < 4) { so_panic("Expansion factor must be at least 4"); return -1; }
    return 0;
}
/* fri.rs:93-103 */
uint64_t so_fri_num_rounds(const so_fri_cfg *c) {
    uint64_t len = c->domain_length, r = 0;
    while (len > c->expansion_factor && 4 * c->num_colinearity_tests < len) { len /= 2; r++; }
    return r;
}
/* fri.rs:57-91: per element one exp, two div (each an xgcd), four mul */
void so_fri_fold_codeword(const so_fri_cfg *c, const uint64_t *cw, size_t len, uint64_t alpha,
                          uint64_t offset, uint64_t omega, uint64_t *out) {
    uint64_t p = c->p, one = 1;
    uint64_t two_inv = so_ff_inv(p, 2);
    size_t half = len / 2;
    for (size_t i = 0; i < half; i++) {
        uint64_t x = so_ff_mul(p, offset, so_ff_exp(p, omega, (uint64_t)i));
        uint64_t a = so_ff_add(p, one, so_ff_div(p, alpha, x));
        uint64_t b = so_ff_sub(p, one, so_ff_div(p, alpha, x));
        uint64_t term = so_ff_add(p, so_ff_mul(p, a, cw[i]), so_ff_mul(p, b, cw[half + i]));
        out[i] = so_ff_mul(p, two_inv, term);
    }
}
/* fri.rs:158-166 */
void so_fri_eval_domain(const so_fri_cfg *c, size_t round, uint64_t *out) {
    size_t size = c->domain_length >> round;
    for (size_t i = 0; i < size; i++)
        out[i] = so_ff_mul(c->p, c->offset, so_ff_exp(c->p, c->omega, ((uint64_t)1 << round) * i));
}
/* fri.rs:168-174: u128 shift-xor accumulator -> effectively the last 8 bytes, big endian */
size_t so_fri_sample_index(const uint8_t *bytes, size_t nbytes, size_t size) {
    u128 acc = 0;
    for (size_t i = 0; i < nbytes; i++) acc = (acc << 8) ^ (u128)bytes[i];
    return (size_t)((uint64_t)acc % size);
}
/* fri.rs:176-213 */
int so_fri_sample_indices(const uint8_t *seed, size_t seed_len, size_t size, size_t reduced_size,
                          size_t number, uint64_t *out) {
    if (!(number <= 2 * reduced_size)) { so_panic("not enough entropy in indices wrt last codeword"); return -1; }
    if (!(number <= reduced_size)) { so_panic("cannot sample more indices than available in last codeword"); return -1; }
    uint64_t *reduced = xmalloc((number + 1) * 8);
    uint8_t *buf = xmalloc(seed_len + 4);
    size_t cnt = 0; uint32_t counter = 0;
    while (cnt < number) {
        memcpy(buf, seed, seed_len);
        for (int b = 0; b < 4; b++) buf[seed_len + b] = (uint8_t)(counter >> (8 * b));
        uint8_t h[32];
        so_hash_from_bytes(buf, seed_len + 4, h);
        size_t index = so_fri_sample_index(h, 32, size);
        size_t ri = index % reduced_size;
        counter += 1;
        int seen = 0;
        for (size_t k = 0; k < cnt; k++) if (reduced[k] == ri) { seen = 1; break; }
        if (!seen) { out[cnt] = index; reduced[cnt] = ri; cnt++; }
    }
    free(reduced); free(buf);
    return 0;
}

static void leaf_hashes(const uint64_t *cw, size_t len, uint8_t *out) {
    for (size_t i = 0; i < len; i++) so_hash_from_field_elements(&cw[i], 1, out + i * 32); /* fri.rs:118-121 */
}
static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

/* fri.rs:105-156.  codewords_out[r]/lens_out[r] receive malloc'd copies (R entries, or 1 if R==0). */
static size_t fri_commit(const so_fri_cfg *c, const uint64_t *initial, size_t len, pstream *ps, so_fs *fs,
                         uint64_t **codewords_out, size_t *lens_out, uint8_t *roots_out, uint64_t *alphas_out) {
    uint64_t p = c->p, omega = c->omega, offset = c->offset;
    uint64_t R = so_fri_num_rounds(c);
    uint64_t *cw = xmalloc(len * 8); memcpy(cw, initial, len * 8);
    size_t ncw = 0;
    for (uint64_t r = 0; r < R; r++) {
        size_t padded = next_pow2(len);
        uint8_t *hashes = xmalloc(padded * 32);
        leaf_hashes(cw, len, hashes);
        memset(hashes + len * 32, 0, (padded - len) * 32);             /* :123-125 */
        uint8_t *nodes = xmalloc((2 * padded - 1) * 32);
        so_merkle_new(hashes, padded, nodes);                          /* :127 */
        const uint8_t *root = nodes + (2 * padded - 2) * 32;
        ps_push_root(ps, root);                                        /* :129 */
        so_fs_absorb(fs, root, 32);                                    /* :131 */
        if (roots_out) memcpy(roots_out + r * 32, root, 32);
        free(hashes); free(nodes);
        if (r == R - 1) break;                                         /* :133-135 */
        uint64_t alpha = so_fs_challenge(fs);                          /* :138 */
        if (alphas_out) alphas_out[r] = alpha;
        codewords_out[ncw] = xmalloc(len * 8); memcpy(codewords_out[ncw], cw, len * 8); lens_out[ncw] = len; ncw++;
        uint64_t *folded = xmalloc((len / 2 + 1) * 8);
        so_fri_fold_codeword(c, cw, len, alpha, offset, omega, folded); /* :143 */
        free(cw); cw = folded; len /= 2;
        omega = so_ff_mul(p, omega, omega);                            /* :146-147 */
        offset = so_ff_mul(p, offset, offset);
    }
    ps_push_fes(ps, cw, len);                                          /* :151 */
    codewords_out[ncw] = cw; lens_out[ncw] = len; ncw++;               /* :153 */
    return ncw;
}

int so_fri_commit_trace(const so_fri_cfg *c, const uint64_t *codeword, size_t len, uint8_t *roots_out,
                        uint64_t *alphas_out, uint64_t *last_codeword_out, size_t *last_len) {
    if (so_fri_new_check(c)) return -1;
    pstream ps; ps_init(&ps);
    so_fs *fs = so_fs_new();
    uint64_t *cws[64]; size_t lens[64];
    size_t n = fri_commit(c, codeword, len, &ps, fs, cws, lens, roots_out, alphas_out);
    memcpy(last_codeword_out, cws[n - 1], lens[n - 1] * 8);
    *last_len = lens[n - 1];
    for (size_t i = 0; i < n; i++) free(cws[i]);
    so_fs_free(fs); ps_free(&ps);
    return g_panic[0] ? -1 : 0;
}

/* fri.rs:215-248 */
static void fri_query(const so_fri_cfg *c, const uint64_t *cur, size_t cur_len, const uint64_t *next,
                      const uint64_t *c_idx, pstream *ps, const uint8_t *cur_nodes, const uint8_t *next_nodes,
                      size_t next_len) {
    size_t half = cur_len / 2, t = c->num_colinearity_tests;
    uint8_t path[64 * 32];
    for (size_t s = 0; s < t; s++) {
        uint64_t triple[3] = {cur[c_idx[s]], cur[c_idx[s] + half], next[c_idx[s]]};
        ps_push_fes(ps, triple, 3);
    }
    for (size_t s = 0; s < t; s++) {
        int d;
        d = so_merkle_open(cur_nodes, cur_len, c_idx[s], path); ps_push_path(ps, path, d < 0 ? 0 : (size_t)d);
        d = so_merkle_open(cur_nodes, cur_len, c_idx[s] + half, path); ps_push_path(ps, path, d < 0 ? 0 : (size_t)d);
        d = so_merkle_open(next_nodes, next_len, c_idx[s], path); ps_push_path(ps, path, d < 0 ? 0 : (size_t)d);
    }
}

/* fri.rs:250-311 */
int so_fri_prove(const so_fri_cfg *c, const uint64_t *codeword, size_t len, uint8_t **proof_out,
                 size_t *proof_len, uint64_t *top_indices_out) {
    *proof_out = NULL; *proof_len = 0;
    if (so_fri_new_check(c)) return -1;
    if (c->domain_length != len) { so_panic("initial codeword length does not match domain length"); return -1; }
    pstream ps; ps_init(&ps);
    so_fs *fs = so_fs_new();
    uint64_t *cws[64]; size_t lens[64];
    size_t ncw = fri_commit(c, codeword, len, &ps, fs, cws, lens, NULL, NULL);
    int rc = 0;
    size_t t = c->num_colinearity_tests;
    size_t sample_size = ncw > 1 ? lens[1] : lens[0];                       /* :266-270 */
    uint8_t seed[32];
    so_hash_from_u64(so_fs_challenge(fs), seed);                            /* :272 */
    uint64_t *top = xmalloc((t + 1) * 8), *idx = xmalloc((t + 1) * 8);
    if (so_fri_sample_indices(seed, 32, sample_size, lens[ncw - 1], t, top)) rc = -1;
    if (rc == 0) {
        memcpy(idx, top, t * 8);
        for (size_t i = 0; i + 1 < ncw; i++) {
            for (size_t s = 0; s < t; s++) idx[s] = idx[s] % (lens[i] / 2);  /* :282-285 */
            /* :288-298 both trees are rebuilt from scratch for every layer */
            uint8_t *ch = xmalloc(lens[i] * 32), *nh = xmalloc(lens[i + 1] * 32);
            leaf_hashes(cws[i], lens[i], ch); leaf_hashes(cws[i + 1], lens[i + 1], nh);
            uint8_t *cn = xmalloc((2 * lens[i] - 1) * 32), *nn = xmalloc((2 * lens[i + 1] - 1) * 32);
            if (so_merkle_new(ch, lens[i], cn) || so_merkle_new(nh, lens[i + 1], nn)) rc = -1;
            else fri_query(c, cws[i], lens[i], cws[i + 1], idx, &ps, cn, nn, lens[i + 1]);
            free(ch); free(nh); free(cn); free(nn);
            if (rc) break;
        }
    }
    if (rc == 0) {
        *proof_out = ps_serialize(&ps, proof_len);
        if (top_indices_out) memcpy(top_indices_out, top, t * 8);
    }
    free(top); free(idx);
    for (size_t i = 0; i < ncw; i++) free(cws[i]);
    so_fs_free(fs); ps_free(&ps);
    return (rc || g_panic[0]) ? -1 : 0;
}

/* fri.rs:507-525 */
static int fri_test_colinearity(uint64_t p, const uint64_t xs[3], const uint64_t ys[3]) {
    uint64_t dy1 = so_ff_sub(p, ys[1], ys[0]), dx1 = so_ff_sub(p, xs[1], xs[0]);
    uint64_t dy2 = so_ff_sub(p, ys[2], ys[0]), dx2 = so_ff_sub(p, xs[2], xs[0]);
    return so_ff_mul(p, dy1, dx2) == so_ff_mul(p, dy2, dx1);
}

/* fri.rs:313-504 */
int so_fri_verify(const so_fri_cfg *c, const uint8_t *proof, size_t proof_len, uint64_t *pv_idx,
                  uint64_t *pv_val, size_t *pv_n) {
    uint64_t p = c->p, omega = c->omega, offset = c->offset;
    size_t t = c->num_colinearity_tests, npv = 0;
    uint64_t R = so_fri_num_rounds(c);
    pstream ps; ps_deserialize(&ps, proof, proof_len);
    so_fs *fs = so_fs_new();
    uint8_t *roots = xmalloc((R + 1) * 32);
    uint64_t *alphas = xmalloc((R + 1) * 8);
    uint64_t *last_domain = NULL, *poly = NULL, *reev = NULL, *top = NULL;
    uint8_t *lh = NULL, *ln = NULL;
    uint64_t *aa = xmalloc((t + 1) * 8), *bb = xmalloc((t + 1) * 8), *cc = xmalloc((t + 1) * 8);
    uint64_t *ci = xmalloc((t + 1) * 8);
    int ok = 0;
    g_reject[0] = 0;
#define REJECT(msg) do { so_reject(msg); goto done; } while (0)

    for (uint64_t r = 0; r < R; r++) {                                        /* :325-334 */
        pobj *o = ps_pop(&ps);
        if (!o || o->tag != PO_ROOT) REJECT("Failed to extract Merkle root");
        memcpy(roots + r * 32, o->hash, 32);
        so_fs_absorb(fs, o->hash, 32);
        alphas[r] = so_fs_challenge(fs);
    }
    pobj *lo = ps_pop(&ps);                                                   /* :337-342 */
    if (!lo || lo->tag != PO_FES) REJECT("Failed to extract last codeword");
    const uint64_t *last = lo->fes; size_t last_len = lo->nfes;
    if (R == 0) REJECT("No FRI roots extracted");                             /* :345-348 */
    lh = xmalloc((last_len + 1) * 32);
    leaf_hashes(last, last_len, lh);
    ln = xmalloc((2 * (last_len + 1)) * 32);
    if (so_merkle_new(lh, last_len, ln)) goto done;                           /* :353 panics on bad length */
    if (memcmp(roots + (R - 1) * 32, ln + (2 * last_len - 2) * 32, 32) != 0) REJECT("last codeword is not well formed");

    size_t degree_bound = last_len / c->expansion_factor;                     /* :360 */
    if (degree_bound == 0) REJECT("last codeword too small");
    size_t degree = degree_bound - 1;
    uint64_t last_omega = omega, last_offset = offset;
    for (uint64_t k = 0; k + 1 < R; k++) { last_omega = so_ff_mul(p, last_omega, last_omega); last_offset = so_ff_mul(p, last_offset, last_offset); }
    last_domain = xmalloc(last_len * 8);
    for (size_t i = 0; i < last_len; i++) last_domain[i] = so_ff_mul(p, last_offset, so_ff_exp(p, last_omega, (uint64_t)i));
    poly = xmalloc((last_len + 2) * 8);
    size_t npoly = so_poly_interpolate_domain(p, last_domain, last, last_len, poly); /* :381 */
    if (g_panic[0]) goto done;
    reev = xmalloc(last_len * 8);
    so_poly_eval_domain(p, poly, npoly, last_domain, last_len, reev);         /* :384-390 */
    for (size_t i = 0; i < last_len; i++) if (reev[i] != last[i]) REJECT("re-evaluated codeword does not match original!");
    if (so_poly_deg(poly, npoly) > (int64_t)degree) REJECT("last codeword does not correspond to polynomial of low enough degree");

    uint8_t seed[32];
    so_hash_from_u64(so_fs_challenge(fs), seed);                              /* :400-405 */
    top = xmalloc((t + 1) * 8);
    if (so_fri_sample_indices(seed, 32, c->domain_length >> 1, c->domain_length >> (R - 1), t, top)) goto done;

    for (uint64_t r = 0; r + 1 < R; r++) {                                    /* :408-502 */
        size_t half = c->domain_length >> (r + 1);
        for (size_t s = 0; s < t; s++) ci[s] = top[s] % half;
        for (size_t s = 0; s < t; s++) {
            pobj *o = ps_pop(&ps);
            if (!o || o->tag != PO_FES) REJECT("Failed to extract triple values");
            if (o->nfes != 3) REJECT("Expected triple of values");
            uint64_t ay = o->fes[0], by = o->fes[1], cy = o->fes[2];
            aa[s] = ay; bb[s] = by; cc[s] = cy;
            if (r == 0 && pv_idx && pv_val) {
                pv_idx[npv] = ci[s]; pv_val[npv] = ay; npv++;
                pv_idx[npv] = ci[s] + half; pv_val[npv] = by; npv++;
            }
            uint64_t xs[3] = {so_ff_mul(p, offset, so_ff_exp(p, omega, ci[s])),
                              so_ff_mul(p, offset, so_ff_exp(p, omega, ci[s] + half)), alphas[r]};
            uint64_t ys[3] = {ay, by, cy};
            if (!fri_test_colinearity(p, xs, ys)) REJECT("colinearity check failure");
        }
        for (size_t i = 0; i < t; i++) {
            uint8_t leaf[32];
            pobj *o = ps_pop(&ps);
            if (!o || o->tag != PO_PATH) REJECT("Failed to extract path for aa");
            so_hash_from_field_elements(&aa[i], 1, leaf);
            if (!so_merkle_verify(leaf, ci[i], o->path, o->npath, roots + r * 32)) REJECT("merkle authentication path verification fails for aa");
            o = ps_pop(&ps);
            if (!o || o->tag != PO_PATH) REJECT("Failed to extract path for bb");
            so_hash_from_field_elements(&bb[i], 1, leaf);
            if (!so_merkle_verify(leaf, ci[i] + half, o->path, o->npath, roots + r * 32)) REJECT("merkle authentication path verification fails for bb");
            o = ps_pop(&ps);
            if (!o || o->tag != PO_PATH) REJECT("Failed to extract path for cc");
            so_hash_from_field_elements(&cc[i], 1, leaf);
            if (!so_merkle_verify(leaf, ci[i], o->path, o->npath, roots + (r + 1) * 32)) REJECT("merkle authentication path verification fails for cc");
        }
        omega = so_ff_mul(p, omega, omega);
        offset = so_ff_mul(p, offset, offset);
    }
    ok = 1;
done:
#undef REJECT
    if (pv_n) *pv_n = npv;
    free(roots); free(alphas); free(last_domain); free(poly); free(reev); free(top);
    free(lh); free(ln); free(aa); free(bb); free(cc); free(ci);
    so_fs_free(fs); ps_free(&ps);
    if (g_panic[0]) return -1;
    return ok;
}

/* ----------------------------------------------- fast CPU NTT restatement */
/* Not in the reference (F1: it has no NTT).  For a geometric domain d[k] = s*w^k,
 * interpolate_domain gives c[j] = s^-j n^-1 sum_k v[k] w^-jk and eval_domain gives
 * E[k] = sum_j c[j] s^j W^jk; these compute exactly those sums with radix-2
 * butterflies.  tests/test_oracle_fast.py proves equality with the O(n^3)/O(Nd)
 * restatements above for every n <= 2^10 it tries. */
static void bitrev_permute(uint64_t *a, size_t n) {
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}
static void ntt_inplace(uint64_t p, uint64_t w, uint64_t *a, size_t n) {
    bitrev_permute(a, n);
    for (size_t len = 2; len <= n; len <<= 1) {
        uint64_t wl = so_ff_exp(p, w, n / len);
        size_t half = len / 2;
        uint64_t *tw = xmalloc(half * 8);
        tw[0] = 1;
        for (size_t k = 1; k < half; k++) tw[k] = so_ff_mul(p, tw[k - 1], wl);
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; k++) {
                uint64_t u = a[i + k], v = so_ff_mul(p, a[i + k + half], tw[k]);
                a[i + k] = so_ff_add(p, u, v);
                a[i + k + half] = so_ff_sub(p, u, v);
            }
        free(tw);
    }
}
void so_fast_intt(uint64_t p, uint64_t omega, uint64_t offset, const uint64_t *vals, size_t n, uint64_t *coeffs) {
    uint64_t *a = xmalloc(n * 8);
    for (size_t i = 0; i < n; i++) a[i] = vals[i] % p;
    ntt_inplace(p, so_ff_inv(p, omega), a, n);
    uint64_t ninv = so_ff_inv(p, (uint64_t)n % p), sinv = so_ff_inv(p, offset), f = ninv;
    for (size_t j = 0; j < n; j++) { coeffs[j] = so_ff_mul(p, a[j], f); f = so_ff_mul(p, f, sinv); }
    free(a);
}
void so_fast_coset_ntt(uint64_t p, uint64_t omega_N, uint64_t offset, const uint64_t *coeffs, size_t nc,
                       size_t N, uint64_t *evals) {
    uint64_t f = 1;
    /* coefficients beyond N wrap onto x^(j mod N) * s^j scaling: callers keep nc <= N */
    for (size_t j = 0; j < N; j++) evals[j] = 0;
    for (size_t j = 0; j < nc; j++) {
        evals[j % N] = so_ff_add(p, evals[j % N], so_ff_mul(p, coeffs[j] % p, f));
        f = so_ff_mul(p, f, offset);
    }
    ntt_inplace(p, omega_N, evals, N);
}
/* out[i] = 2^-1 ((1 + a/x_i) c[i] + (1 - a/x_i) c[i+h]), x_i = offset*omega^i, with running x_i^-1 */
void so_fast_fold(uint64_t p, const uint64_t *cw, size_t len, uint64_t alpha, uint64_t offset,
                  uint64_t omega, uint64_t *out) {
    size_t half = len / 2;
    uint64_t two_inv = so_ff_inv(p, 2), xinv = so_ff_inv(p, offset), winv = so_ff_inv(p, omega);
    uint64_t a = alpha % p;
    for (size_t i = 0; i < half; i++) {
        uint64_t ax = so_ff_mul(p, a, xinv);
        uint64_t s = so_ff_add(p, cw[i], cw[half + i]), d = so_ff_sub(p, cw[i], cw[half + i]);
        out[i] = so_ff_mul(p, two_inv, so_ff_add(p, s, so_ff_mul(p, ax, d)));
        xinv = so_ff_mul(p, xinv, winv);
    }
}
