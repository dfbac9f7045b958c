"""GPU: builds the C++ host-mirror test (tests/cpp/fri_mirror_test.cpp over include/stark_mi.hpp)
against libstarkmi.so and runs it as a child process."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_replays_reference_tests(oracle):
    import stark_rs_amd as s
    s.build()
    lib_dir = os.path.join(ROOT, "stark_rs_amd", "build")
    ora_dir = os.path.join(ROOT, "oracle", "build")
    exe = os.path.join(lib_dir, "fri_mirror_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fri_mirror_test.cpp"),
                           f"-L{lib_dir}", "-lstarkmi", f"-L{ora_dir}", "-lstark_oracle", "-L/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{ora_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL PASSED" in out.stdout and out.stdout.count("verified") == 4
