"""ctypes loader for libstarkmi.so (the C ABI of include/stark_mi.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable the import of
an Engine fails loudly (StarkMiError).  Nothing here imports oracle/.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMI_EMU_LIB: another build of the CPU emulator (tests only; the sanitizer build of `make asan`)
EMU_PATH = os.environ.get("SMI_EMU_LIB") or os.path.join(_HERE, "build", "libstarkmi_emu.so")
LIB_PATH = os.environ.get("SMI_LIB") or os.path.join(_HERE, "build", "libstarkmi.so")   # SMI_LIB: tuning builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "stark_mi.h")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class StarkMiError(RuntimeError):
    """A non-zero status from the C ABI.  str() is the reference's panic message for the
    codes that mirror one (smi_status_string), so tests can match on it like
    `#[should_panic(expected = ...)]`."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 56), ("launches", C.c_uint32), ("total_ms", C.c_double), ("alg_bytes", C.c_double),
                ("alg_mixes", C.c_double)]


class StarkCfg(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("log_blowup", C.c_uint32), ("n_cols", C.c_uint32), ("row_leaves", C.c_uint32),
                ("trace_offset", C.c_uint64), ("lde_offset", C.c_uint64), ("num_colinearity_tests", C.c_uint64),
                ("open_columns", C.c_uint64)]


class FriCfg(C.Structure):
    _fields_ = [("omega", C.c_uint64), ("offset", C.c_uint64), ("domain_length", C.c_uint64),
                ("expansion_factor", C.c_uint64), ("num_colinearity_tests", C.c_uint64)]


def build(force=False):
    """Compile libstarkmi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + [HEADER]
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)   # a sanitizer runtime preloaded into the test process must not reach the compilers
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-j4"], env=env)
    return LIB_PATH


def declared_symbols():
    """Every function name include/stark_mi.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StarkMiError(-100, f"{LIB_PATH} is missing: run `make -C stark_rs_amd` (hipcc, gfx950). "
                                 "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same SONAME as the
    # system one).  If torch is loaded after us the process ends up with two runtimes and torch
    # then sees no GPU; importing it first makes our DT_NEEDED resolve to the copy it loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    sz = C.c_size_t
    i32 = C.c_int
    sig = {
        "smi_status_string": (C.c_char_p, [i32]),
        "smi_last_error": (C.c_char_p, [vp]),
        "smi_version": (C.c_char_p, []),
        "smi_ctx_create": (i32, [C.c_uint64, C.c_uint64, i32, C.POINTER(vp)]),
        "smi_ctx_destroy": (None, [vp]),
        "smi_ctx_set_stream": (i32, [vp, vp]),
        "smi_ctx_sync": (i32, [vp]),
        "smi_ctx_profile": (i32, [vp, i32]),
        "smi_ctx_profile_only": (i32, [vp, C.c_char_p]),
        "smi_ctx_copy_probe": (i32, [vp, i32]),
        "smi_ctx_mix_probe": (i32, [vp, C.c_uint32, C.POINTER(C.c_double)]),
        "smi_ctx_lde_two_pass": (i32, [vp, i32]),
        "smi_ctx_profile_read": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "smi_ctx_modulus": (C.c_uint64, [vp]),
        "smi_ctx_two_adicity": (C.c_uint32, [vp]),
        "smi_prim_nth_root": (i32, [vp, C.c_uint64, u64p]),
        "smi_ff_inv": (i32, [vp, C.c_uint64, u64p]),
        "smi_ff_exp": (i32, [vp, C.c_uint64, C.c_uint64, u64p]),
        "smi_ff_mul": (i32, [vp, C.c_uint64, C.c_uint64, u64p]),
        "smi_intt": (i32, [vp, vp, vp, C.c_uint32, C.c_uint64]),
        "smi_coset_ntt": (i32, [vp, vp, sz, vp, C.c_uint32, C.c_uint64]),
        "smi_poly_scale": (i32, [vp, vp, sz, C.c_uint64, vp]),
        "smi_poly_mul": (i32, [vp, vp, sz, vp, sz, vp, C.POINTER(sz)]),
        "smi_poly_div": (i32, [vp, vp, sz, vp, sz, vp, C.POINTER(sz), vp, C.POINTER(sz)]),
        "smi_domain_is_geometric": (i32, [vp, vp, sz, u64p]),
        "smi_lde": (i32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp]),
        "smi_trace_pack": (i32, [vp, vp, sz, sz, vp]),
        "smi_hash_leaves": (i32, [vp, vp, sz, vp]),
        "smi_hash_combine_pairs": (i32, [vp, vp, sz, vp]),
        "smi_hash_bytes": (i32, [vp, C.c_char_p, sz, vp]),
        "smi_hash_bytes_batch": (i32, [vp, C.c_char_p, sz, sz, vp]),
        "smi_dev_hash_bytes": (i32, [vp, vp, sz, vp]),
        "smi_merkle_commit": (i32, [vp, vp, sz, vp]),
        "smi_merkle_new": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "smi_merkle_from_codeword": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "smi_merkle_root": (i32, [vp, vp, vp]),
        "smi_merkle_open": (i32, [vp, vp, sz, vp, C.POINTER(sz)]),
        "smi_merkle_level": (i32, [vp, vp, C.c_uint32, vp, C.POINTER(sz)]),
        "smi_merkle_verify_batch": (i32, [vp, vp, vp, vp, sz, sz, vp, vp]),
        "smi_merkle_num_leaves": (sz, [vp]),
        "smi_merkle_free": (None, [vp]),
        "smi_fri_check": (i32, [vp, C.POINTER(FriCfg)]),
        "smi_fri_num_rounds": (i32, [C.POINTER(FriCfg), u64p]),
        "smi_fri_fold": (i32, [vp, vp, sz, C.c_uint64, C.c_uint64, C.c_uint64, vp]),
        "smi_fri_commit": (i32, [vp, C.POINTER(FriCfg), vp, sz, vp, vp, vp, C.POINTER(sz), C.POINTER(vp)]),
        "smi_fri_prove": (i32, [vp, C.POINTER(FriCfg), vp, sz, C.POINTER(vp), C.POINTER(sz), vp]),
        "smi_fri_verify": (i32, [vp, C.POINTER(FriCfg), C.c_char_p, sz, C.POINTER(i32), vp, vp, C.POINTER(sz)]),
        "smi_stark_verify": (i32, [vp, C.POINTER(StarkCfg), vp, C.c_char_p, sz, C.POINTER(i32)]),
        "smi_fri_run_num_codewords": (i32, [vp, C.POINTER(sz)]),
        "smi_fri_run_codeword": (i32, [vp, sz, vp, C.POINTER(sz)]),
        "smi_fri_run_open": (i32, [vp, sz, sz, vp, C.POINTER(sz)]),
        "smi_fri_run_free": (None, [vp]),
        "smi_free": (None, [vp]),
        "smi_dev_alloc": (i32, [vp, sz, C.POINTER(vp)]),
        "smi_dev_free": (i32, [vp, vp]),
        "smi_dev_upload_u64": (i32, [vp, vp, sz, vp, i32]),
        "smi_dev_download_u64": (i32, [vp, vp, sz, vp]),
        "smi_dev_ntt": (i32, [vp, vp, vp, C.c_uint32, sz, C.c_uint32, sz, sz, i32, C.c_uint64, C.c_uint64]),
        "smi_dev_lde": (i32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp]),
        "smi_dev_hash_leaves": (i32, [vp, vp, sz, vp]),
        "smi_dev_merkle_build": (i32, [vp, vp, sz, vp]),
        "smi_dev_merkle_build_rows": (i32, [vp, vp, C.c_uint32, sz, sz, vp]),
        "smi_dev_merkle_from_digests": (i32, [vp, sz, vp]),
        "smi_dev_fri_fold": (i32, [vp, vp, sz, vp, C.c_uint64, C.c_uint64, vp]),
        "smi_dev_fri_fold_shard": (i32, [vp, vp, vp, sz, sz, sz, vp, C.c_uint64, C.c_uint64, vp]),
        "smi_dev_fri_prove": (i32, [vp, C.POINTER(FriCfg), vp, sz, C.POINTER(vp), C.POINTER(sz), vp, C.POINTER(vp)]),
        "smi_dev_combine_columns": (i32, [vp, vp, C.c_uint32, sz, sz, vp, vp]),
        "smi_dev_stark_prove": (i32, [vp, C.POINTER(StarkCfg), vp, vp, C.POINTER(vp), C.POINTER(sz), vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def status_string(status):
    return lib().smi_status_string(status).decode()


def check(status, ctx=None):
    if status == 0:
        return
    msg = status_string(status)
    if ctx is not None and (status <= -50):
        detail = lib().smi_last_error(ctx).decode()
        if detail:
            msg = f"{msg}: {detail}"
    raise StarkMiError(status, msg)
