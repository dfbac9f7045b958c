#!/bin/bash
# Development tool (GPU box): where the waves of the prove's hash kernels spend their cycles (SQ counters) and the effective
# clock (GRBM_GUI_ACTIVE / 8 / duration), plus the same for the bare mix loop (tools/ubench_mix).   bash tools/profile_stalls_prove.sh <tag>
set -e
TAG=${1:-run}
OUT=gpurun_out/stalls_prove_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for target in prove ubench; do
  if [ $target = prove ]; then CMD="python3 tools/prove_time.py 22 profiled"; export REPS=3; else CMD="tools/ubench_mix"; fi
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES -d $OUT/sq_$target -o sq -- $CMD > /dev/null 2> $OUT/sq_$target.err
  rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_WAVES -d $OUT/g_$target -o g -- $CMD > /dev/null 2> $OUT/g_$target.err || echo "second counter set failed for $target"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def load(sub):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows
def dur(sub):
    d = collections.defaultdict(list)
    for fn in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            d[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d
for target in ("prove", "ubench"):
    for sub in (f"sq_{target}", f"g_{target}"):
        rows, d = load(sub), dur(sub)
        for k in sorted(rows):
            if not re.search(r"merkle|void k<|mix_probe", k): continue
            m = {c: sum(v) / len(v) for c, v in rows[k].items()}
            us = sum(d[k]) / len(d[k]) / 1e3
            name = re.sub(r"\(.*", "", k).replace("void ", "")[:48]
            if sub.startswith("sq"):
                wc = m.get("SQ_WAVE_CYCLES", 0) or 1
                print(f"{name:48s} {us:8.0f} us  " + "  ".join(f"{c[3:].lower()} {v / wc:.3f}" for c, v in sorted(m.items()) if c not in ("SQ_WAVE_CYCLES", "SQ_WAVES")) + f"  wave_cycles/wave {wc / max(m.get('SQ_WAVES', 1), 1):.0f}")
            else:
                w = m.get("SQ_WAVES", 0) or 1
                clk = m.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3
                print(f"{name:48s} {us:8.0f} us  clock {clk:.2f} GHz  per wave: " + "  ".join(f"{c[3:].lower()} {v / w:.0f}" for c, v in sorted(m.items()) if c.startswith("SQ_") and c != "SQ_WAVES") + f"  waves {w:.0f}")
PY
