#!/usr/bin/env python3
"""Development tool: condense the rocprofv3 outputs of tools/profile.sh into the small files kept
under profiles/ (kernel stats CSVs, PMC traffic per launch, the bench line of the stats run).

HBM traffic per launch = FETCH_SIZE * 2 + WRITE_SIZE (both reported in KB): on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, "HBM"), WRITE_SIZE is exact for
streaming stores."""
import csv
import glob
import json
import os
import re
import shutil
import sys

KIND = {"0": "first", "1": "mid", "2": "last"}


def bench_name(sym):
    """rocprofv3 kernel symbol -> the name bench.py / smi_ctx_profile use."""
    m = re.search(r"ntt_pass(_cols)?_kernel<(\d+), (\d+), (\d+), \d+(?:, \w+)?>", sym)   # _cols: the columns of a tile per workgroup
    if m:
        return f"ntt_pass{m.group(1) or ''}_kernel<{m.group(2)},{m.group(3)},{KIND[m.group(4)]}>"
    m = re.search(r"lde_a_kernel<(\d+), \d+>", sym)
    if m:
        return f"lde_a_kernel<{m.group(1)}>"
    if "lde_b_kernel" in sym:
        return "lde_b_kernel"
    m = re.search(r"merkle_sub_kernel<(true|false)(?:, (\d+), (true|false)(?:, (\d+))?)?>", sym)
    if m:
        kind = "leaves" if m.group(1) == "true" else "digests"
        if m.group(3) == "true":
            kind = "row leaves"
        src = {None: "", "0": "", "1": " fold", "2": " combine"}.get(m.group(4), "")     # LeafSrc kind: leaves computed by the kernel
        return f"merkle_sub_kernel<{kind}>" + (f" K={m.group(2)}" if m.group(2) and m.group(2) != "0" else "") + src
    m = re.search(r"(\w+_kernel)", sym)
    return m.group(1) if m else sym


def counters(run_dir):
    """{bench kernel name: {counter: [values per dispatch]}}"""
    out = {}
    for path in glob.glob(os.path.join(run_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = bench_name(row["Kernel_Name"])
            out.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return out


def summarise(src, suffix=""):
    fetch, write, valu = (counters(os.path.join(src, d + suffix)) for d in ("fetch", "write", "valu"))
    summary = {}
    for k in sorted(set(fetch) | set(write)):
        if not (k.startswith("ntt_pass_") or k.startswith("lde_")):
            continue
        f = fetch.get(k, {}).get("FETCH_SIZE", [])
        w = write.get(k, {}).get("WRITE_SIZE", [])
        e = {"launches_profiled": len(f)}
        if f and w:
            e["FETCH_SIZE_KB_avg"] = sum(f) / len(f)
            e["WRITE_SIZE_KB_avg"] = sum(w) / len(w)
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024.0
        v = valu.get(k, {})
        if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = sum(v["SQ_INSTS_VALU"]) / sum(v["SQ_WAVES"])
        summary[k] = e
    return summary


NOTE = ("rocprofv3 --pmc in separate passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVES) over "
        "`bench.py --no-extras`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
        "requests at 64 B); KB units; averages over all launches of the kernel in the run")


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(dst, exist_ok=True)
    if len(sys.argv) > 3 and sys.argv[3] == "prove":          # VALU instructions per wave of the prove's kernels
        valu = counters(os.path.join(src, "prove_valu"))
        out = {}
        for k, v in sorted(valu.items()):
            if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES") and sum(v["SQ_WAVES"]):
                out[k] = {"launches_profiled": len(v["SQ_WAVES"]), "valu_insts_per_wave": sum(v["SQ_INSTS_VALU"]) / sum(v["SQ_WAVES"])}
        json.dump({"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES over tools/kbench.py prove:22:3:4 (2^22 x 4 trace, blowup 8)", **out},
                  open(os.path.join(dst, f"{tag}_prove_pmc.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))
        return
    if len(sys.argv) > 3 and sys.argv[3] == "two_pass":       # the opt-in two-pass extension, SMI_LDE_TWO_PASS=1
        for path in glob.glob(os.path.join(src, "stats2", "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(path, os.path.join(dst, f"{tag}_lde_two_pass_kernel_stats.csv"))
        summary = summarise(src, "2")
        json.dump({"note": NOTE + "; SMI_LDE_TWO_PASS=1", **summary}, open(os.path.join(dst, f"{tag}_lde_two_pass_pmc.json"), "w"), indent=1)
        traffic_path = os.path.join(dst, "pmc_traffic.json")
        traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
        traffic.update({k: v for k, v in summary.items() if k.startswith("lde_")})
        json.dump(traffic, open(traffic_path, "w"), indent=1)
        bj = os.path.join(src, "bench_stats2.json")
        if os.path.exists(bj):
            lines = [l for l in open(bj) if l.startswith("{")]
            if lines:
                open(os.path.join(dst, f"{tag}_two_pass_bench_under_rocprof.json"), "w").write(lines[-1])
        print(json.dumps(summary, indent=1))
        return
    for sub, name in (("stats", "lde"), ("prove", "prove")):
        for path in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(path, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    summary = summarise(src)
    json.dump({"note": NOTE, **summary}, open(os.path.join(dst, f"{tag}_lde_pmc.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    bj = os.path.join(src, "bench_stats.json")
    if os.path.exists(bj):
        lines = [l for l in open(bj) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(lines[-1])
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
