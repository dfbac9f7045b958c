"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/stark_mi.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os

import pytest


def test_library_builds_and_exports_every_declared_symbol():
    import stark_rs_amd as s
    path = s.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = s.declared_symbols()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_status_strings_are_the_reference_panic_messages():
    from stark_rs_amd import _lib
    L = _lib.lib()
    want = {
        -1: "no inverse",                                                  # src/ff.rs:171
        -2: "no division by zero",                                         # src/ff.rs:182
        -3: "n must be a power of two",                                    # src/ff.rs:217
        -4: "n > 2^23 not supported by this modulus",                      # src/ff.rs:218
        -5: "Cannot create tree from empty leaves",                        # src/merkle.rs:12
        -6: "Number of leaves must be power of 2",                         # src/merkle.rs:13-16
        -7: "Index out of bounds",                                         # src/merkle.rs:68
        -8: "Domain length must be power of 2",                            # src/fri.rs:37-40
        -9: "Expansion factor must be power of 2",                         # src/fri.rs:41-44
        -10: "Expansion factor must be at least 4",                        # src/fri.rs:45
        -11: "initial codeword length does not match domain length",       # src/fri.rs:256-260
        -12: "not enough entropy in indices wrt last codeword",            # src/fri.rs:183-186
    }
    for code, msg in want.items():
        assert L.smi_status_string(code).decode() == msg


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a usable device the engine must raise, never compute on the CPU."""
    import torch
    import stark_rs_amd as s
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(s.StarkMiError, match="no CPU fallback"):
        s.Engine()


def test_num_rounds_host_logic():
    import ctypes as C
    from stark_rs_amd import _lib
    L = _lib.lib()
    for (n, e, t, want) in [(32, 4, 2, 2), (64, 4, 3, 3), (128, 4, 4, 3), (256, 8, 5, 4), (1 << 23, 8, 32, 16),
                            (1 << 25, 8, 32, 18), (8, 8, 1, 0)]:
        cfg = _lib.FriCfg(1, 1, n, e, t)
        r = C.c_uint64()
        assert L.smi_fri_num_rounds(C.byref(cfg), C.byref(r)) == 0
        assert r.value == want                                             # src/fri.rs:93-103, SURVEY a10


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under stark_rs_amd/ may reference it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stark_rs_amd")
    for dirpath, _, files in os.walk(root):
        if "build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "stark_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
