"""CPU: runs the NTT / hash kernels' per-thread phase functions (csrc/ntt_core.h,
csrc/hash_core.h -- the same code the HIP kernels execute) one "thread" at a time through
the emulator library and checks them against the oracle.  This is how index / twiddle /
planning logic is validated in the GPU-less build container; the emulator is test
infrastructure and is never loaded by the product."""
import ctypes as C
import os

import numpy as np
import pytest

P, G = 998244353, 3
P2, G2 = 469762049, 3
u32p = C.POINTER(C.c_uint32)


@pytest.fixture(scope="module")
def emu():
    import stark_rs_amd as s
    s.build()
    from stark_rs_amd._lib import EMU_PATH as path
    L = C.CDLL(path)
    L.emu_ntt.argtypes = [C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                          C.c_int, C.c_uint64, C.c_uint64]
    return L


def _ntt(emu, p, g, x, L, n_in, inverse, offset, post=1, batch=1, in_stride=None):
    x = np.ascontiguousarray(x, dtype=np.uint32)
    out = np.zeros(batch << L, dtype=np.uint32)
    rc = emu.emu_ntt(p, g, x.ctypes.data_as(u32p), out.ctypes.data_as(u32p), L, n_in, batch,
                     (len(x) // batch) if in_stride is None else in_stride, 1 << L, inverse, offset, post)
    assert rc == 0
    return out.astype(np.uint64)


@pytest.mark.parametrize("p,g", [(P, G), (P2, G2)])
@pytest.mark.parametrize("L", [0, 1, 4, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_emulated_ntt_every_plan(emu, oracle, p, g, L):
    o = oracle
    n = 1 << L
    w = o.ff_prim_nth_root_g(n, p, g)
    vals = o.splitmix64(L, n) % np.uint64(p)
    assert np.array_equal(_ntt(emu, p, g, vals, L, n, 1, 3), o.fast_intt(vals, w, 3, p))
    nin = max(1, n // 8)
    assert np.array_equal(_ntt(emu, p, g, vals[:nin], L, nin, 0, 3), o.fast_coset_ntt(vals[:nin], n, w, 3, p))


def test_emulated_ntt_three_pass_and_batch(emu, oracle):
    o = oracle
    L = 21
    n = 1 << L
    w = o.ff_prim_nth_root(n)
    vals = o.splitmix64(5, n) % np.uint64(P)
    assert np.array_equal(_ntt(emu, P, G, vals, L, n, 0, 1), o.fast_coset_ntt(vals, n, w, 1))
    # batch of 3 columns of 2^14, fused post-scale (the LDE's coset shift)
    L = 14
    n = 1 << L
    w = o.ff_prim_nth_root(n)
    cols = o.splitmix64(6, 3 * n) % np.uint64(P)
    got = _ntt(emu, P, G, cols, L, n, 1, 1, post=3, batch=3).reshape(3, n)
    for c in range(3):
        coeffs = o.fast_intt(cols[c * n:(c + 1) * n], w, 1)
        assert np.array_equal(got[c], np.array(o.poly_scale(coeffs, 3), dtype=np.uint64))


def test_emulated_hash_core(emu, oracle):
    o = oracle
    rng = np.random.default_rng(1)
    v = np.concatenate([np.array([0, 1, 5, P - 1, 0xFFFFFFFF, 255, 256], dtype=np.uint32),
                        rng.integers(0, 2 ** 32, 500, dtype=np.uint32)])
    out = np.zeros((len(v), 32), dtype=np.uint8)
    emu.emu_leaf_hash(v.ctypes.data_as(C.c_void_p), C.c_size_t(len(v)), out.ctypes.data_as(C.c_void_p))
    for i in range(len(v)):
        assert bytes(out[i]) == o.hash_from_field_elements([int(v[i])])
    pairs = rng.integers(0, 256, (200, 64), dtype=np.uint8)
    out = np.zeros((200, 32), dtype=np.uint8)
    emu.emu_node_hash(pairs.ctypes.data_as(C.c_void_p), C.c_size_t(200), out.ctypes.data_as(C.c_void_p))
    for i in range(200):
        assert bytes(out[i]) == o.hash_combine(bytes(pairs[i, :32]), bytes(pairs[i, 32:]))
    for n in list(range(0, 70)) + [100, 255, 256, 1000]:
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        ob = (C.c_uint8 * 32)()
        emu.emu_hash_bytes(m, C.c_size_t(n), ob)
        assert bytes(ob) == o.hash_from_bytes(m)


def test_emulated_node_hash_over_a_quad_of_lanes(emu, oracle):
    """hash_quad.h (the upper Merkle levels of merkle_top_kernel: one hash spread over four lanes, the
    ring add as a local prefix plus a 4-lane scan over quad_perm moves) stepped in lockstep on the CPU:
    Hash::combine of random, all-zero, all-ones and repeated-byte children."""
    o = oracle
    rng = np.random.default_rng(9)
    pairs = rng.integers(0, 256, (300, 64), dtype=np.uint8)
    pairs[0] = 0
    pairs[1] = 255
    pairs[2, :32] = 0
    pairs[3, 32:] = 255
    for k in range(4, 20):
        pairs[k] = (k * 37) & 255
    out = np.zeros((len(pairs), 32), dtype=np.uint8)
    emu.emu_node_hash_quad(pairs.ctypes.data_as(C.c_void_p), C.c_size_t(len(pairs)), out.ctypes.data_as(C.c_void_p))
    for i in range(len(pairs)):
        assert bytes(out[i]) == o.hash_combine(bytes(pairs[i, :32]), bytes(pairs[i, 32:])), i


def test_emulated_hex_node_hash(emu, oracle):
    """hash_hex.h -- the node hash of the narrowest Merkle levels, one state word per lane of a row of sixteen (S-box per
    lane, linear mix over quad_perm moves, ring add as a row scan, absorb recurrence as five rotate-by-seven stages)
    stepped in lockstep on the CPU: Hash::combine of random, all-zero, all-ones, repeated-byte and single-byte children."""
    o = oracle
    rng = np.random.default_rng(10)
    pairs = rng.integers(0, 256, (400, 64), dtype=np.uint8)
    pairs[0] = 0
    pairs[1] = 255
    pairs[2, :32] = 0
    pairs[3, 32:] = 255
    for k in range(4, 20):
        pairs[k] = (k * 37) & 255
    for k in range(64):   # one non-zero byte at every position: each step of the absorb recurrence on its own
        pairs[20 + k] = 0
        pairs[20 + k, k] = 0x80 | k
    out = np.zeros((len(pairs), 32), dtype=np.uint8)
    emu.emu_node_hash_hex(pairs.ctypes.data_as(C.c_void_p), C.c_size_t(len(pairs)), out.ctypes.data_as(C.c_void_p))
    for i in range(len(pairs)):
        assert bytes(out[i]) == o.hash_combine(bytes(pairs[i, :32]), bytes(pairs[i, 32:])), i


def test_emulated_hex_fiat_shamir_round(emu, oracle):
    """hashx::fs_absorb / fs_challenge -- the Fiat-Shamir round at the end of a tree's last launch, over a row of sixteen
    lanes: a transcript of 12 roots against the single-lane form (state bytes and challenges) and against the oracle's
    FiatShamir (src/fiat_shamir.rs:9-25) run on the same roots."""
    rng = np.random.default_rng(11)
    st_a = (C.c_uint32 * 16)(*[int(p_) * 0x00010001 for p_ in [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53]])
    st_b = (C.c_uint32 * 16)(*st_a)
    fs = oracle.FiatShamir()
    for k in range(12):
        root = bytes(rng.integers(0, 256, 32, dtype=np.uint8)) if k else bytes(32)
        a, b = C.c_uint64(0), C.c_uint64(0)
        emu.emu_fs_round_hex(st_a, root, C.byref(a))
        emu.emu_fs_round(st_b, root, C.byref(b))
        fs.absorb(root)
        assert a.value == b.value == fs.challenge(), k
        # the stored words agree on every meaningful bit (bits 0..7 of either 16-bit lane)
        assert [x & 0x00FF00FF for x in st_a] == [x & 0x00FF00FF for x in st_b], k


@pytest.mark.parametrize("L,plan", [
    (18, "9.3,9.3"), (18, "9.5,9.4"), (18, "10.4,8.4"), (18, "6.6,6.6,6.6"), (19, "7.6,6.6,6.6"), (20, "10.2,10.2"),
    (20, "10.4,10.3"), (20, "7.5,7.5,6.6"), (20, "8.6,6.6,6.6"), (21, "8.5,7.6,6.6"), (21, "9.4,6.6,6.6"),
])
def test_emulated_ntt_explicit_tile_shapes(emu, oracle, L, plan):
    """Every instantiated (logr, logw) tile shape, selected through the tuning override."""
    o = oracle
    os.environ[f"SMI_NTT_PLAN_{L}"] = plan
    try:
        n = 1 << L
        w = o.ff_prim_nth_root(n)
        vals = o.splitmix64(L, n) % np.uint64(P)
        assert np.array_equal(_ntt(emu, P, G, vals, L, n, 1, 3), o.fast_intt(vals, w, 3))
        assert np.array_equal(_ntt(emu, P, G, vals[: n // 8], L, n // 8, 0, 3), o.fast_coset_ntt(vals[: n // 8], n, w, 3))
    finally:
        del os.environ[f"SMI_NTT_PLAN_{L}"]


@pytest.mark.parametrize("p,g", [(P, G), (P2, G2)])
@pytest.mark.parametrize("L", [13, 16, 18])
def test_emulated_ntt_extreme_values_stay_in_range(emu, oracle, p, g, L):
    """The lazy butterflies track a static bound m*p per register (ntt_core.h) and fold only where
    the next sum would pass 2^32.  Constant and alternating inputs of p-1 drive every sum path to
    its bound; the emulator build asserts each claimed bound (SMI_EMU_CHECKS) and the result must
    still be the oracle's."""
    o = oracle
    n = 1 << L
    w = o.ff_prim_nth_root_g(n, p, g)
    pats = [np.full(n, p - 1, dtype=np.uint64),
            np.where(np.arange(n) % 2 == 0, p - 1, 0).astype(np.uint64),
            np.where((np.arange(n) >> 4) % 2 == 0, p - 1, 1).astype(np.uint64)]
    for vals in pats:
        assert np.array_equal(_ntt(emu, p, g, vals, L, n, 0, 1), o.fast_coset_ntt(vals, n, w, 1, p))   # no coset scale
        assert np.array_equal(_ntt(emu, p, g, vals, L, n, 1, 1), o.fast_intt(vals, w, 1, p))
        nin = n // 8
        assert np.array_equal(_ntt(emu, p, g, vals[:nin], L, nin, 0, 3), o.fast_coset_ntt(vals[:nin], n, w, 3, p))


@pytest.mark.parametrize("W", [0, 1, 2, 3, 4, 5, 7, 8, 9])
def test_emulated_row_leaf_hash(emu, oracle, W):
    """Hash::from_field_elements(&row) for rows of W residues (hash_core.h row_hash / row_hash2):
    full chunks of four elements, short last chunks, the empty row."""
    o = oracle
    rng = np.random.default_rng(W)
    n = 21
    v = rng.integers(0, P, (n, max(W, 1)), dtype=np.int64).astype(np.uint32)
    if W:
        v[0, :] = P - 1
        v[1, :] = 0
    out = np.zeros((n, 32), dtype=np.uint8)
    emu.emu_row_hash(v.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_int(W), out.ctypes.data_as(C.c_void_p))
    for i in range(n):
        assert bytes(out[i]) == o.hash_from_field_elements([int(x) for x in v[i, :W]])


def test_emulated_ntt_batched_4096_point_columns_use_pass_kernels(emu, oracle):
    """64 or more columns of 4096 points go through two passes of 64-point lines instead of the
    single-workgroup kernel (planner, ntt_host.h)."""
    o = oracle
    L, batch = 12, 64
    n = 1 << L
    w = o.ff_prim_nth_root(n)
    cols = o.splitmix64(11, batch * n) % np.uint64(P)
    got = _ntt(emu, P, G, cols, L, n, 1, 3, batch=batch).reshape(batch, n)
    for c in (0, 1, 37, 63):
        assert np.array_equal(got[c], o.fast_intt(cols[c * n:(c + 1) * n], w, 3))
    got = _ntt(emu, P, G, cols, L, n, 0, 5, batch=batch).reshape(batch, n)
    for c in (0, 63):
        assert np.array_equal(got[c], o.fast_coset_ntt(cols[c * n:(c + 1) * n], n, w, 5))


def _lde2(emu, p, g, coef, L, beta, batch=1):
    coef = np.ascontiguousarray(coef, dtype=np.uint32)
    out = np.zeros(batch << (L + beta), dtype=np.uint32)
    emu.emu_lde2.argtypes = [C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32]
    rc = emu.emu_lde2(p, g, coef.ctypes.data_as(u32p), out.ctypes.data_as(u32p), L, beta, batch)
    assert rc == 0
    return out.astype(np.uint64)


# two-pass low-degree extension (csrc/lde_core.h): every line size of pass A (2^10, 2^11, 2^12) and
# every coset/line split of pass B (blowup 2..16), both lazy ranges, against Polynomial::eval_domain's
# fast restatement on the blowup subgroup (src/univariate/eval.rs:16-21)
@pytest.mark.parametrize("p,g,L,beta", [(P, G, 20, 1), (P, G, 20, 2), (P, G, 20, 3), (P2, G2, 20, 4), (P2, G2, 21, 3),
                                        (P2, G2, 22, 3), (P, G, 22, 1)])
def test_emulated_two_pass_lde(emu, oracle, p, g, L, beta):
    o = oracle
    n, N = 1 << L, 1 << (L + beta)
    W = o.ff_prim_nth_root_g(N, p, g)
    coef = o.splitmix64(31 * L + beta, n) % np.uint64(p)
    assert np.array_equal(_lde2(emu, p, g, coef, L, beta), o.fast_coset_ntt(coef, N, W, 1, p))


def test_emulated_two_pass_lde_extremes_and_batch(emu, oracle):
    """constant p-1 / alternating inputs drive every lazy sum to its bound (the emulator asserts the
    claimed ranges); a batch of two columns checks the column strides."""
    o = oracle
    L, beta = 20, 3
    n, N = 1 << L, 1 << (L + beta)
    for p, g in ((P, G), (P2, G2)):
        W = o.ff_prim_nth_root_g(N, p, g)
        a = np.full(n, p - 1, dtype=np.uint64)
        b = a.copy()
        b[1::2] = 0
        got = _lde2(emu, p, g, np.concatenate([a, b]), L, beta, batch=2).reshape(2, N)
        assert np.array_equal(got[0], o.fast_coset_ntt(a, N, W, 1, p))
        assert np.array_equal(got[1], o.fast_coset_ntt(b, N, W, 1, p))


def test_field_setup_rejects_composite_moduli(emu):
    """every inverse in the engine is a Fermat power, so field_setup (csrc/tables.h) must refuse an odd
    composite with enough two-adicity (20481 = 3 * 6827 = 5 * 2^12 + 1) -- and still take both primes."""
    x = np.arange(16, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    args = (x.ctypes.data_as(u32p), out.ctypes.data_as(u32p), 4, 16, 1, 16, 16, 0, 1, 1)
    assert emu.emu_ntt(20481, 3, *args) == -1
    assert emu.emu_ntt(P, G, *args) == 0 and emu.emu_ntt(P2, G2, *args) == 0


def test_emulated_ntt_deferred_first_twiddle(emu, oracle):
    """three-pass plans (2^21 x 2 columns) with the first pass's inter-pass twiddle applied by the second
    pass as it loads (NTT_TW_SKIP / NTT_TW_IN of csrc/ntt_core.h): same results, zero-padded (an
    extension) and full, forward and inverse."""
    o = oracle
    L = 21
    n = 1 << L
    emu.emu_set_defer_tw(1)
    try:
        # second prime again with the columns of a tile in one workgroup: the deferred twiddles become 16 per-thread
        # input multipliers derived once (NttPass::in_mul, ntt_pass_cols_kernel<..., TWIN>)
        for p, g, share in ((P, G, 1), (P2, G2, 1), (P2, G2, 2)):
            emu.emu_set_share_cols(share)
            w = o.ff_prim_nth_root_g(n, p, g)
            a, b = o.splitmix64(7, n) % np.uint64(p), np.full(n, p - 1, dtype=np.uint64)
            both = np.concatenate([a, b])
            got = _ntt(emu, p, g, both, L, n, 0, 3, batch=2).reshape(2, n)
            assert np.array_equal(got[0], o.fast_coset_ntt(a, n, w, 3, p)) and np.array_equal(got[1], o.fast_coset_ntt(b, n, w, 3, p))
            got = _ntt(emu, p, g, both, L, n // 8, 0, 1, batch=2, in_stride=n).reshape(2, n)       # zero-padded: an extension
            assert np.array_equal(got[0], o.fast_coset_ntt(a[:n // 8], n, w, 1, p))
            assert np.array_equal(got[1], o.fast_coset_ntt(b[:n // 8], n, w, 1, p))
            got = _ntt(emu, p, g, both, L, n, 1, 1, batch=2).reshape(2, n)
            assert np.array_equal(got[0], o.fast_intt(a, w, 1, p)) and np.array_equal(got[1], o.fast_intt(b, w, 1, p))
    finally:
        emu.emu_set_defer_tw(2)
        emu.emu_set_share_cols(1)


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("p,g", [(P, G), (P2, G2)])
@pytest.mark.parametrize("L", [13, 14, 15, 16, 17, 18, 19, 20])
def test_emulated_ntt_last_pass_from_load_registers(emu, oracle, p, g, L, direct):
    """NTT_LAST_DIRECT (csrc/ntt_core.h): the last pass runs its first radix-16 step on the registers its loads landed
    in (lanes along a line, lines dealt to the lanes rotated) instead of transposing through LDS first -- every plan,
    forward zero-padded and inverse with its output scale, one column and three (the column-sharing last pass); direct = 0
    is the transposing load the knob SMI_NTT_LAST_DIRECT=0 keeps."""
    o = oracle
    n = 1 << L
    w = o.ff_prim_nth_root_g(n, p, g)
    vals = o.splitmix64(100 + L, n) % np.uint64(p)
    emu.emu_set_last_direct(direct)
    emu.emu_set_share_cols(2)
    try:
        assert np.array_equal(_ntt(emu, p, g, vals, L, n, 1, 3), o.fast_intt(vals, w, 3, p))
        nin = max(1, n // 8)
        assert np.array_equal(_ntt(emu, p, g, vals[:nin], L, nin, 0, 3), o.fast_coset_ntt(vals[:nin], n, w, 3, p))
        if L <= 17:
            cols = [vals, np.full(n, p - 1, dtype=np.uint64), o.splitmix64(7 * L, n) % np.uint64(p)]
            got = _ntt(emu, p, g, np.concatenate(cols), L, n, 1, 5, batch=3).reshape(3, n)
            for c in range(3):
                assert np.array_equal(got[c], o.fast_intt(cols[c], w, 5, p))
    finally:
        emu.emu_set_last_direct(1)
        emu.emu_set_share_cols(1)


@pytest.mark.parametrize("L,batch", [(14, 3), (17, 6), (21, 2)])
def test_emulated_ntt_columns_of_a_tile_in_one_workgroup(emu, oracle, L, batch):
    """ntt_pass_cols_kernel (csrc/ntt.hip): with more than one column, a middle pass (and a last pass with an
    output scale) takes every column of its tile in one workgroup and derives the thread's 16 output multipliers
    once (NttPass::out_mul), four columns per workgroup.  Same results as one workgroup per (tile, column) and as the oracle: forward on a
    coset, zero-padded (an extension), inverse with its n^-1 / offset^-j output scale."""
    o = oracle
    n = 1 << L
    for p, g in ((P, G), (P2, G2)):
        w = o.ff_prim_nth_root_g(n, p, g)
        cols = [o.splitmix64(31 + c, n) % np.uint64(p) for c in range(batch)]
        cols[-1] = np.full(n, p - 1, dtype=np.uint64)
        flat = np.concatenate(cols)
        runs = {}
        for share in (2, 0):          # 2: whenever the pass has multipliers (the launch-size threshold of the rule aside)
            emu.emu_set_share_cols(share)
            try:
                runs[share] = (_ntt(emu, p, g, flat, L, n, 0, 3, batch=batch).reshape(batch, n),
                               _ntt(emu, p, g, flat, L, n // 4, 0, 5, batch=batch, in_stride=n).reshape(batch, n),
                               _ntt(emu, p, g, flat, L, n, 1, 7, batch=batch).reshape(batch, n))
            finally:
                emu.emu_set_share_cols(1)
        for a, b in zip(runs[2], runs[0]):
            assert np.array_equal(a, b)
        for c in range(batch):
            assert np.array_equal(runs[2][0][c], o.fast_coset_ntt(cols[c], n, w, 3, p))
            assert np.array_equal(runs[2][1][c], o.fast_coset_ntt(cols[c][:n // 4], n, w, 5, p))
            assert np.array_equal(runs[2][2][c], o.fast_intt(cols[c], w, 7, p))


@pytest.mark.parametrize("geo,lay", [(2, 3), (2, 4), (3, 1), (3, 3), (1, 3), (1, 4)])
def test_emulated_two_pass_lde_layout_and_tile_geometry_knobs(emu, oracle, geo, lay):
    """the intermediate's block size (lay_kq) and pass B's split of its 16 lines into cosets x adjacent k1
    (geo_rq) are independent tuning knobs (SMI_LDE_LAYOUT / SMI_LDE_GEO): every combination is the same map"""
    o = oracle
    L, beta = 20, 3
    n, N = 1 << L, 1 << (L + beta)
    W = o.ff_prim_nth_root_g(N, P2, G2)
    coef = o.splitmix64(77, n) % np.uint64(P2)
    emu.emu_lde2_knobs(geo, lay)
    try:
        assert np.array_equal(_lde2(emu, P2, G2, coef, L, beta), o.fast_coset_ntt(coef, N, W, 1, P2))
    finally:
        emu.emu_lde2_knobs(-1, -1)
