#!/bin/bash
# Development tool (GPU box): where the waves of the extension's pass kernels spend their cycles (SQ counters, one
# pass of 8) and the effective clock (GRBM_GUI_ACTIVE / 8 / duration).   bash tools/profile_stalls.sh <tag>
set -e
TAG=${1:-run}
OUT=gpurun_out/stalls_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-extras --in-loop-only --steps 100 --warmup 50"   # sustained regime (r03)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/sq -o sq -- $BENCH > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES -d $OUT/grbm -o g -- $BENCH > /dev/null 2> $OUT/grbm.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def load(sub):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in f:
        for r in csv.DictReader(open(fn)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows
def dur(sub):
    f = glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True)
    d = collections.defaultdict(list)
    for fn in f:
        for r in csv.DictReader(open(fn)):
            d[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d
for sub in ("sq", "grbm"):
    rows, d = load(sub), dur(sub)
    for k in sorted(rows):
        if "ntt_pass" not in k: continue
        m = {c: sum(v) / len(v) for c, v in rows[k].items()}
        n = len(next(iter(rows[k].values())))
        if n < 5: continue
        us = sum(d[k]) / len(d[k]) / 1e3
        name = re.sub(r"void |\(PassArgs\)", "", k)
        if sub == "sq":
            wc = m["SQ_WAVE_CYCLES"]
            print(f"{name:44s} {us:6.0f} us  wait_any {m['SQ_WAIT_ANY']/wc:.2f} wait_inst {m['SQ_WAIT_INST_ANY']/wc:.2f} active {m['SQ_ACTIVE_INST_ANY']/wc:.2f} valu {m['SQ_ACTIVE_INST_VALU']/wc:.2f} lds {m['SQ_ACTIVE_INST_LDS']/wc:.2f} wait_lds {m['SQ_WAIT_INST_LDS']/wc:.2f} bank_conf {m['SQ_LDS_BANK_CONFLICT']/wc:.3f}")
        else:
            w = m["SQ_WAVES"]
            print(f"{name:44s} {us:6.0f} us  clock {m['GRBM_GUI_ACTIVE']/8/us/1e3:.2f} GHz  per wave: valu {m['SQ_INSTS_VALU']/w:.0f} lds {m['SQ_INSTS_LDS']/w:.0f} vmem_rd {m['SQ_INSTS_VMEM_RD']/w:.0f} vmem_wr {m['SQ_INSTS_VMEM_WR']/w:.0f} salu {m['SQ_INSTS_SALU']/w:.0f}  waves {w:.0f}")
PY
