import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/stark_oracle.c) -- the checker, never the product."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


def column_openings_bytes(o, lde, top, N):
    """smi_stark_cfg.open_columns restated with the oracle: per test s, the rows of the W extended columns at
    a = top[s] mod N/2 and b = a + N/2 (FieldElements, src/stream.rs:48-53), then per (s, c) MerkleTree::open
    (src/merkle.rs:67-80) of column c's tree at a and b (MerklePath, src/stream.rs:54-59)."""
    u64 = lambda v: int(v).to_bytes(8, "little")
    W, half = len(lde), N // 2
    trees = [o.merkle_new(o.leaf_hashes(col)) for col in lde]
    out = bytearray()
    for s in top:
        a = s % half
        for i in (a, a + half):
            out += b"\x02" + u64(W) + b"".join(u64(col[i]) for col in lde)
    for s in top:
        a = s % half
        for c in range(W):
            for i in (a, a + half):
                path = o.merkle_open(trees[c], N, i)
                out += b"\x03" + u64(len(path)) + b"".join(bytes(d) for d in path)
    return bytes(out)
