#!/bin/bash
# Development tool (CPU only, never on the GPU box): AddressSanitizer + UBSan builds of the CPU emulator
# (the kernels' per-thread code, the multi-GPU round loop, the proof parser) and of the oracle, and the
# CPU tests that drive them.   bash tools/run_sanitizers.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
make -s -C stark_rs_amd all asan      # the plain build first: nothing may start a compiler under the preloaded runtime
make -s -C oracle all asan
ASAN_SO=$(gcc -print-file-name=libasan.so)
UBSAN_SO=$(gcc -print-file-name=libubsan.so)
export SMI_EMU_LIB=$PWD/stark_rs_amd/build/asan/libstarkmi_emu.so
export SMI_ORACLE_LIB=$PWD/oracle/build/asan/libstark_oracle.so
# python itself is not instrumented: leak checking would only report the interpreter's own arenas
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD="$ASAN_SO $UBSAN_SO" python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider \
    tests/test_proof_parse.py tests/test_emu_kernels.py tests/test_mgpu_gloo.py tests/test_oracle_kats.py tests/test_oracle_fast.py "$@"
