#!/usr/bin/env python3
"""Development tool: timeline of the LAST prove in a rocprofv3 --kernel-trace CSV (tools/prove_time.py under the
profiler): per-kernel busy time, and the idle gaps between consecutive kernels on the stream.
    python3 tools/trace_gaps.py gpurun_out/<dir>/prove_kernel_trace.csv [n_proves_in_trace]"""
import csv
import re
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda s: re.sub(r"\(.*", "", s).replace("void ", "")
# a prove starts at its first ntt_pass launch after a fri/column kernel: split on the small inverse first pass
starts = [i for i, r in enumerate(rows) if "ntt_pass_kernel<8, 5, 0" in r["Kernel_Name"]]
if not starts:
    raise SystemExit("no prove found")
a = starts[-1]
seg = rows[a:]
t0 = int(seg[0]["Start_Timestamp"])
busy, gaps, count = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end, prev_name, total_gap = None, None, 0.0
for r in seg:
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    busy[n] += (e - s) / 1e3
    count[n] += 1
    if prev_end is not None and s > prev_end:
        gaps[f"{prev_name} -> {n}"] += (s - prev_end) / 1e3
        total_gap += (s - prev_end) / 1e3
    prev_end, prev_name = max(prev_end or 0, e), n
span = (prev_end - t0) / 1e3
print(f"last prove: {len(seg)} launches, span {span:.1f} us, busy {sum(busy.values()):.1f} us, idle between kernels {total_gap:.1f} us")
for n, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  {n[:70]:70s} x{count[n]:<3d} {v:9.1f} us")
print("largest gap classes:")
for n, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {n[:110]:110s} {v:8.1f} us")
