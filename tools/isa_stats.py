#!/usr/bin/env python3
"""Development tool: per-thread VALU instruction counts of a kernel from the compiler's ISA listing
(hipcc -save-temps), by issue class (tools/ubench_valu.hip: "fast" = 2 cycles per wave, "slow" = 4), with
the bodies of backward-branch loops weighted by a trip count (the hash kernels' closing-mix loops run 8 times).

    python3 tools/isa_stats.py hash-hip-amdgcn-amd-amdhsa-gfx950.s 'merkle_sub_kernelILb1ELi2ELb0' [trips=8]
"""
import re
import sys
from collections import Counter

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_bitop3_b32", "v_fma_f32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_mul_lo_u16", "v_add_u16", "v_mov_b32"}


def kernel_body(text, pat):
    m = re.search(r"^(_Z\w*" + pat + r"\w*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    if not m:
        raise SystemExit(f"no kernel matching {pat}")
    return m.group(1), m.group(2).splitlines()


def stats(lines, trips):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    weight = [1] * len(lines)
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            for k in range(labels[m.group(1)], i + 1):
                weight[k] *= trips
    dyn, cls = Counter(), Counter()
    for l, w in zip(lines, weight):
        m = re.match(r"\s+((?:v|s|ds|global|buffer|scratch)_\w+)", l)
        if not m:
            continue
        op = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", m.group(1))
        dyn[op] += w
        if op.startswith("v_"):
            src_sgpr = bool(re.search(r",\s*s\d+|,\s*0x[0-9a-f]{3,}", l))     # an SGPR or literal source demotes a fast op
            cls["fast" if (op in FAST and not src_sgpr) else "slow"] += w
    return dyn, cls


def main():
    text = open(sys.argv[1]).read()
    trips = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    name, lines = kernel_body(text, sys.argv[2])
    dyn, cls = stats(lines, trips)
    valu = sum(v for k, v in dyn.items() if k.startswith("v_"))
    print(name)
    print(f"  VALU per thread (loops x{trips}): {valu}   fast {cls['fast']}  slow {cls['slow']}   issue cycles per wave ~ {2 * cls['fast'] + 4 * cls['slow']}")
    print("  scratch:", sum(v for k, v in dyn.items() if k.startswith("scratch")), " lds:", sum(v for k, v in dyn.items() if k.startswith("ds_")),
          " global:", sum(v for k, v in dyn.items() if k.startswith("global")))
    print("  ", ", ".join(f"{k} {v}" for k, v in dyn.most_common(14)))


if __name__ == "__main__":
    main()
