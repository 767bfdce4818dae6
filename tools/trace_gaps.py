#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV.
usage: trace_gaps.py <kernel_trace.csv> [last_n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_\w+|nccl\w+|rccl\w+|__amd_\w+|vectorized\w+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:32]
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:8.1f} us"
    print(f"{name:28s} dur {(e - s) / 1e3:8.1f} us  {gap}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
    prev_end = e
