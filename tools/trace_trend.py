#!/usr/bin/env python3
"""Development tool: per-call durations of the NTT pass kernels over time from a rocprofv3 --kernel-trace CSV of the
bench loop (explains the in-loop spread: the first tens of milliseconds after the GPU leaves idle run slower).
    python3 tools/trace_trend.py gpurun_out/<dir>/lde_kernel_trace.csv [bucket=50]"""
import csv
import re
import statistics
import sys

bucket = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "ntt_pass" in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"])
by = {}
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    by.setdefault(n, []).append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("kernel durations over the whole trace (us):")
for n, v in by.items():
    d = [x[1] for x in v]
    print(f"  {n:45s} n={len(d):4d} mean {statistics.mean(d):7.1f} median {statistics.median(d):7.1f} min {min(d):7.1f} max {max(d):7.1f} sd {statistics.pstdev(d):5.1f}")
dom = max(by.items(), key=lambda kv: sum(x[1] for x in kv[1]))
print(f"\n{dom[0]}: mean duration per {bucket} consecutive calls")
for i in range(0, len(dom[1]), bucket):
    b = dom[1][i:i + bucket]
    print(f"  calls {i:5d}-{i + len(b) - 1:5d}  t = {b[0][0]:8.1f} ms   mean {statistics.mean(x[1] for x in b):6.1f}  min {min(x[1] for x in b):6.1f}  max {max(x[1] for x in b):6.1f}")
first = min(by, key=lambda n: by[n][0][0])
st = [x[0] * 1e3 for x in by[first]]
d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
print("\nstep time (start of a step's first kernel to the next one's), mean per bucket (us):")
for i in range(0, len(d), bucket):
    b = d[i:i + bucket]
    print(f"  steps {i:5d}-{i + len(b) - 1:5d}  {statistics.mean(b):7.1f}")
