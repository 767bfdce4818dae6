#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: counter values and duration per dispatch, as the MEAN over all dispatches
of the kernel and for its LAST dispatch of every pass (bench.py's last launch of the batch kernel is the timed pipeline of K
frames; the earlier ones are its set-up and warm-up launches).

usage: pmc_summary.py <pmc_dir> [traffic.json]
With a second argument also writes the per-kernel numbers, with the byte conversions MI355X_MICROARCH.md (HBM / rocprofv3)
prescribes for gfx950: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in units of 1024 B; FETCH_SIZE tallies 128-B requests at
64 B, so fabric reads = 2 x FETCH_SIZE; WRITE_SIZE is taken as reported.  (The gather pattern of k_render is not the streaming
pattern that correction was calibrated on, so its figure is an upper estimate; tools/pmc_calib gives the same counters on a
known streaming read and a known record gather.)  TCC_HIT / TCC_MISS / TCC_REQ count L2 requests of one 128-byte line each
(calibration: profiles/r3/pmc_calib.txt).
"""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> [(file, dispatch id, value)]
dur = collections.defaultdict(dict)                                       # kernel -> (file, dispatch) -> us
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?|calib_\w+|__amd_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        d = int(r["Dispatch_Id"])
        vals[k][r["Counter_Name"]].append((f, d, float(r["Counter_Value"])))
        dur[k][(f, d)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for k in sorted(vals):
    ds = list(dur[k].values())
    last_of = {}                                                            # file -> last dispatch id of this kernel
    for (f, d) in dur[k]:
        last_of[f] = max(last_of.get(f, -1), d)
    last_d = [dur[k][(f, d)] for f, d in last_of.items()]
    print(f"{k}   dispatches={len(ds)} over {len(last_of)} passes  mean_duration_us={sum(ds) / len(ds):.3f}  last_dispatch_duration_us={sum(last_d) / len(last_d):.3f}")
    o = out.setdefault(k, {"mean": {}, "last": {}, "mean_duration_us": sum(ds) / len(ds), "last_duration_us": sum(last_d) / len(last_d)})
    for c, lst in sorted(vals[k].items()):
        mean = sum(v for _, _, v in lst) / len(lst)
        last = [v for f, d, v in lst if d == last_of[f]]
        lastm = sum(last) / len(last)
        o["mean"][c], o["last"][c] = mean, lastm
        print(f"   {c:32s} n={len(lst):4d} mean={mean:.6g}  last={lastm:.6g}")
    for which in ("mean", "last"):
        c = o[which]
        der = o.setdefault("derived_" + which, {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            der["fabric_bytes"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            der["l2_hit_bytes"], der["l2_miss_bytes"] = int(c["TCC_HIT_sum"] * 128), int(c["TCC_MISS_sum"] * 128)
            der["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 4)
        if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU", 0) > 0:
            der["valu_lane_util"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launch = GRBM_GUI_ACTIVE / 8 (calib_stream: 169 us x 2.4 GHz)
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if "TA_BUSY_avr" in c and cyc > 0:
            der["ta_busy"] = round(c["TA_BUSY_avr"] / cyc, 4)                     # busy cycles of the average TA instance / launch cycles
        if "SQ_INSTS_VALU" in c and cyc > 0:
            der["valu_busy"] = round(4.0 * c["SQ_INSTS_VALU"] / (1024.0 * cyc), 4)  # a wave64 VALU instruction holds its SIMD16 for 4 cycles; 256 CUs x 4 SIMDs
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
            der["wave_wait_share"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        if "TCP_TCC_READ_REQ_LATENCY_sum" in c and c.get("TCP_TCC_READ_REQ_sum", 0) > 0:
            der["l1_miss_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
    print("   => last dispatch:", json.dumps(o["derived_last"]))
    # kept for scripts/collect_profiles.py's build-traffic table (mean over dispatches)
    if "fabric_bytes" in o["derived_mean"]:
        o["hbm_bytes_per_launch"] = o["derived_mean"]["fabric_bytes"]
        o["fetch_size_kb_raw"], o["write_size_kb_raw"] = o["mean"]["FETCH_SIZE"], o["mean"]["WRITE_SIZE"]
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
