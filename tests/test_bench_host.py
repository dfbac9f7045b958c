"""CPU: the host-side arithmetic of bench.py that needs no GPU -- the prove's roofline block from per-kernel records,
the launcher's refusal to mislabel a run, and the synthetic-input generator against the oracle's."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_splitmix64_is_the_surveys_generator(oracle):
    b = _bench()
    for seed in (1, 2, 0x5354524B00, 2 ** 63 + 5):
        assert np.array_equal(b.splitmix64(seed, 1000), oracle.splitmix64(seed, 1000))


def test_hash_roofline_block_from_kernel_records():
    b = _bench()
    N, W = 1 << 25, 4
    # one prove: the leaf kernel of the column trees and of the first FRI rounds, the digest levels, the chunk kernel, an NTT pass
    kernels = {
        "merkle_sub_kernel<leaves>": {"launches": 7, "total_ms": 8.4, "alg_bytes": 1.2e10, "alg_mixes": 3.3e9},
        "merkle_sub_kernel<digests>": {"launches": 9, "total_ms": 1.45, "alg_bytes": 1.0e9, "alg_mixes": 4.7e8},
        "merkle_top_kernel": {"launches": 34, "total_ms": 0.8, "alg_bytes": 1e8, "alg_mixes": 4.9e7},
        "ntt_pass_kernel<8,5,last>": {"launches": 1, "total_ms": 0.23, "alg_bytes": 1.07e9, "alg_mixes": 0.0},
    }
    stage = {"lde": 0.72, "commit": 6.6, "combine": 0.02, "fri": 4.25}
    r = b.hash_roofline(kernels, 4.7e11, stage, N, W)
    assert r["kernel"] == "merkle_sub_kernel<leaves>" and r["bound"] == "valu" and r["unit"] == "mix_state/s"
    scale = sum(stage.values()) / sum(k["total_ms"] for k in kernels.values())           # the plain prove's clock
    assert abs(r["kernel_ms_per_prove"] - 8.4 * max(1.0, scale)) < 1e-9
    assert abs(r["frac"] - 3.3e9 / (r["kernel_ms_per_prove"] * 1e-3) / 4.7e11) < 1e-12
    assert abs(r["hbm"]["frac"] - 1.2e10 / (r["kernel_ms_per_prove"] * 1e-3) / 1e9 / b.HBM_PEAK_GBS) < 1e-12
    # N*9 + (N-1)*10 mix_state per tree, ~68 N bytes per tree
    assert r["commit"]["mixes"] == W * (19.0 * N - 10.0)
    assert abs(r["commit"]["hbm_frac"] - W * 68.0 * N / 6.6e-3 / 1e9 / b.HBM_PEAK_GBS) < 1e-12
    assert abs(r["whole_prove"]["mixes"] - (3.3e9 + 4.7e8 + 4.9e7)) < 1
    assert set(r["kernels"]) == {"merkle_sub_kernel<leaves>", "merkle_sub_kernel<digests>", "merkle_top_kernel"}
    json.dumps(r)                                                                          # the block goes into the JSON line
    assert b.hash_roofline({"ntt": {"launches": 1, "total_ms": 1.0, "alg_bytes": 1.0, "alg_mixes": 0.0}}, 4.7e11, stage, N, W) is None
    assert b.hash_roofline(kernels, 0.0, stage, N, W) is None


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "refusing to mislabel" in out.stderr
