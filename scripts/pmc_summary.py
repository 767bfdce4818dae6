#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value and duration per dispatch."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["~duration_us(" + r["Counter_Name"] + " pass)"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:44s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
