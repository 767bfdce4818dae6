#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value and duration per dispatch.

usage: pmc_summary.py <pmc_dir> [traffic.json]
With a second argument also writes the HBM traffic per launch of every kernel, corrected as
MI355X_MICROARCH.md (HBM / rocprofv3) prescribes for gfx950: rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in units of 1024 B; FETCH_SIZE tallies 128-B requests at 64 B, so reads =
2 x FETCH_SIZE; WRITE_SIZE is taken as reported.  (The gather pattern of k_render is not the
streaming pattern that correction was calibrated on, so its figure is an upper estimate.)
"""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
traffic = {}
for k in sorted(acc):
    d = dur[k]
    print(f"{k}   dispatches/pass~{len(d) // max(1, len(glob.glob(root + '/*/')))}  mean_duration_us={sum(d) / len(d):.3f}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:24s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
    if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        fs = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
        ws = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
        traffic[k] = {"fetch_size_kb_raw": fs, "write_size_kb_raw": ws,
                      "hbm_bytes_per_launch": int((2.0 * fs + ws) * 1024)}
        print(f"   => HBM bytes/launch (2*FETCH+WRITE, KB->B) = {traffic[k]['hbm_bytes_per_launch']}")
    if "SQ_THREAD_CYCLES_VALU" in acc[k] and "SQ_ACTIVE_INST_VALU" in acc[k]:
        # the `lanes` pass: active lanes per issued VALU instruction / 64 (SQ_ACTIVE_INST_VALU is also collected by the `sq` pass: mean of both)
        tc = sum(acc[k]["SQ_THREAD_CYCLES_VALU"]) / len(acc[k]["SQ_THREAD_CYCLES_VALU"])
        ai = sum(acc[k]["SQ_ACTIVE_INST_VALU"]) / len(acc[k]["SQ_ACTIVE_INST_VALU"])
        if ai > 0:
            traffic.setdefault(k, {})["valu_lane_util"] = round(tc / (64.0 * ai), 4)
            print(f"   => VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {tc / (64.0 * ai):.4f}")
if len(sys.argv) > 2:
    json.dump(traffic, open(sys.argv[2], "w"), indent=1, sort_keys=True)
