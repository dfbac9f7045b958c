"""CPU: the proof parser smi_fri_verify / smi_stark_verify run over bytes they did not write
(csrc/proof_parse.h = ProofStream::deserialize, reference src/stream.rs:66-168), compiled into the CPU
emulator library and driven with well-formed, truncated, mutated and random byte strings.  Checked against a
pure-Python restatement of the reference's loop and against the oracle's own deserializer (through
so_fri_verify, which must not crash on any of them).  Under tools/run_sanitizers.sh the same test runs on the
AddressSanitizer + UBSan build: an out-of-bounds read on a forged length field would abort it."""
import ctypes as C

import numpy as np
import pytest

P = 998244353


@pytest.fixture(scope="module")
def emu():
    import stark_rs_amd as s
    from stark_rs_amd._lib import EMU_PATH
    s.build()
    L = C.CDLL(EMU_PATH)
    L.emu_proof_parse.restype = C.c_size_t
    L.emu_proof_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.c_size_t, C.POINTER(C.c_size_t)]
    return L


def py_deserialize(b: bytes, max_objs=None):
    """src/stream.rs:66-168, loop for loop -> ([(tag, count, payload offset)], end offset)"""
    out, i, n = [], 0, len(b)
    while i < n and (max_objs is None or len(out) < max_objs):
        tag = b[i]
        i += 1
        if tag == 0:
            if i + 32 <= n:
                out.append((0, 1, i))
                i += 32
        elif tag == 1:
            if i + 8 <= n:
                out.append((1, 1, i))
                i += 8
        elif tag in (2, 3):
            if i + 8 <= n:
                ln = int.from_bytes(b[i:i + 8], "little")
                i += 8
                w = 8 if tag == 2 else 32
                take = min(ln, (n - i) // w)
                out.append((tag, take, i))
                i += take * w
        else:
            i -= 1
            break
    return out, i


def parse(emu, b: bytes, max_objs=2 ** 62):
    cap = len(b) + 2
    tags = (C.c_int32 * cap)()
    counts = (C.c_uint64 * cap)()
    offs = (C.c_uint64 * cap)()
    end = C.c_size_t()
    # an exact-size heap copy, so that the sanitizer build sees any read past the end
    buf = C.create_string_buffer(b, len(b)) if b else None
    k = emu.emu_proof_parse(buf, len(b), max_objs, tags, counts, offs, cap, C.byref(end))
    return [(tags[i], counts[i], offs[i]) for i in range(k)], end.value


def _real_proof(o):
    n, exp, t, offset = 256, 8, 5, 17
    omega = o.ff_prim_nth_root(n)
    cw = o.fast_coset_ntt(o.splitmix64(5, n // exp) % np.uint64(P), n, omega, offset)
    cfg = o.fri_cfg(omega, offset, n, exp, t)
    proof, _ = o.fri_prove(cfg, cw)
    return cfg, proof


def test_parser_equals_the_reference_loop_on_well_formed_and_damaged_proofs(emu, oracle):
    o = oracle
    cfg, proof = _real_proof(o)
    objs, end = parse(emu, proof)
    assert (objs, end) == py_deserialize(proof) and end == len(proof)
    R = o.fri_num_rounds(cfg)
    assert [x[0] for x in objs[:R + 1]] == [0] * R + [2]
    rng = np.random.default_rng(3)
    cases = [proof[:k] for k in list(range(0, 80)) + [len(proof) // 2, len(proof) - 1]]
    for _ in range(300):                                   # byte flips, including tags and length fields
        bad = bytearray(proof)
        for pos in rng.choice(len(bad), size=int(rng.integers(1, 4)), replace=False):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(bad))
    for b in cases:
        assert parse(emu, b) == py_deserialize(b), len(b)
        assert parse(emu, b, 7) == py_deserialize(b, 7)
        try:                                               # the oracle's deserializer survives them too: a verdict or a
            o.fri_verify(cfg, b)                           # reference panic (raised as OraclePanic), never a crash
        except o.OraclePanic:
            pass


def test_parser_clamps_forged_lengths_and_stops_at_unknown_tags(emu):
    u64 = lambda v: int(v).to_bytes(8, "little")
    cases = [
        b"",
        b"\x00",                                           # a root tag with nothing behind it
        b"\x00" + b"\xaa" * 31,                            # one byte short: dropped, its bytes re-read as tags
        b"\x02" + u64(2 ** 64 - 1),                        # length far beyond the buffer: zero elements
        b"\x02" + u64(2 ** 64 - 1) + b"\x01" * 17,         # ... two whole elements, one stray byte
        b"\x03" + u64(2 ** 61) + b"\x07" * 70,             # 2^61 * 32 overflows 64 bits if multiplied first
        b"\x03" + u64(3) + b"\x05" * 96 + b"\x09rest",     # unknown tag 9 ends the stream, offset reported
        b"\x01" + b"\x11" * 7,                             # short FieldElement
        b"\x02" + u64(1)[:5],                              # short length field
    ]
    rng = np.random.default_rng(9)
    for _ in range(2000):
        n = int(rng.integers(0, 200))
        b = bytearray(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        for k in range(0, n, 11):                          # make tags 0..3 frequent
            b[k] = int(rng.integers(0, 5))
        cases.append(bytes(b))
    for b in cases:
        got = parse(emu, b)
        assert got == py_deserialize(b), b[:24]
        for tag, count, off in got[0]:
            w = {0: 32, 1: 8, 2: 8, 3: 32}[tag]
            assert off + count * w <= len(b)               # every object lies inside the buffer
