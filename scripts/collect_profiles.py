#!/usr/bin/env python3
"""Copy what scripts/refresh_profiles.sh measured (gpurun_out/refresh/) into profiles/<round>/ (default r3) and rebuild
profiles/traffic.json, the per-frame PMC figures bench.py's `roofline` reads back.
usage: collect_profiles.py [round-dir-name]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r3"
src, dst = os.path.join(ROOT, "gpurun_out", "refresh"), os.path.join(ROOT, "profiles", RND)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    line = open(f).read().strip().splitlines()
    if not line:
        print("EMPTY", f)
        continue
    json.loads(line[-1])
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
shutil.copy(os.path.join(src, "ks", "kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "project_scaling_*.txt")):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for f in ("pmc_summary.txt", "pmc_calib.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
shutil.copy(os.path.join(src, "gpu_tests.log"), os.path.join(dst, "gpu_tests.log"))      # the -v log: every test by name
allk = json.load(open(os.path.join(src, "traffic_all.json")))
K = 20                                         # bench.py --steps 20: the LAST dispatch of the batch kernels in every pass is the timed pipeline of 20 frames
# the batch pipeline's traversal kernel: the whole-pixel kernel (WIDE64, no counters, textured, KIND_PIXEL, batch) - or, with
# RTCORE_TWO_PHASE=1, KIND_TRAVERSE + KIND_SHADE
trav = [k for k in allk if k.startswith(("k_render<2, false, false, 4, true", "k_render<0, false, false, 4, true"))]   # (LAYOUT 2 = WIDE64 records behind 32-bit offsets)
shade = [k for k in allk if k.startswith(("k_render<2, false, true, 5, true", "k_render<0, false, true, 5, true"))]
if not trav:
    trav = ([k for k in allk if k.startswith("k_render<2, false, true, 6, true")] or [k for k in allk if k.startswith("k_render<0, false, true, 6, true")]
            or [k for k in allk if k.startswith("k_render<0, false, true, 0, true")])   # KIND_PIXEL_NOGI (primary / shadow batches), KIND_PIXEL
    shade = []
assert len(trav) == 1 and len(shade) <= 1, list(allk)


def per_frame(name):
    d = allk[name]["derived_last"]
    return {"l2_hit_bytes_per_frame": d["l2_hit_bytes"] / K, "l2_miss_bytes_per_frame": d["l2_miss_bytes"] / K, "fabric_bytes_per_frame": d["fabric_bytes"] / K,
            "l2_hit_rate": d.get("l2_hit_rate"), "valu_lane_util": d.get("valu_lane_util"), "ta_busy": d.get("ta_busy"), "valu_busy": d.get("valu_busy"),
            "wave_wait_share": d.get("wave_wait_share"), "l1_miss_latency_cycles": d.get("l1_miss_latency_cycles"),
            "launch_us_under_rocprofv3": allk[name]["last_duration_us"], "kernel": name}


t = {"workload": {"scene": "s1m", "width": 3840, "height": 2160, "mode": "shadow", "layout": "wide64", "spp": 32, "textured": True,
                  "frames_per_pipeline": 32, "camera_path": "rotY 1 degree per frame"},
     "source": f"profiles/{RND}/pmc_summary.txt: rocprofv3 --pmc passes of `python3 bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5` "
               "(scripts/pmc_passes.sh), the LAST dispatch of each batch kernel in every pass = the timed pipeline of 20 frames, divided by 20",
     "conversion": "L2 bytes = TCC_HIT_sum / TCC_MISS_sum x 128 B (one request = one 128-byte line: profiles/" + RND + "/pmc_calib.txt, 1 GiB streamed = "
                   "8.389 M requests); fabric bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM / rocprofv3: gfx950 FETCH_SIZE reports "
                   "half of the bytes read; Infinity-Cache hits are counted in it, so it is an upper bound on HBM bytes)",
     "k_render": dict(per_frame(trav[0]), **({"shade": per_frame(shade[0])} if shade else {})),
     "all_kernels_fabric_bytes_per_launch": {k: v["derived_mean"]["fabric_bytes"] for k, v in sorted(allk.items()) if "fabric_bytes" in v.get("derived_mean", {})}}
json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
b = json.loads(open(os.path.join(dst, "bench_n1.json")).read().strip().splitlines()[-1])
print("bench_n1:", b["value"], b["unit"], b["ms_per_step"], "ms; roofline", b["roofline"].get("bound"), b["roofline"]["frac"], "kernel_ms", b["roofline"]["kernel_ms"])

# build: kernel trace + stage times, and the fabric traffic of the build kernels (same PMC passes: bench.py builds 5 times)
for f in ("build_kernel_stats.csv", "build.txt"):
    if os.path.exists(os.path.join(src, "build", f)):
        shutil.copy(os.path.join(src, "build", f), os.path.join(dst, f))
for f in sorted(os.listdir(src)):
    if f.endswith(".txt") and f not in ("pmc.txt", "ks.txt", "pmc_summary.txt", "pmc_calib.txt", "calib.txt"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for f in ("rehearsal_n2.json", "rehearsal_n4.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
build_kernels = [k for k in allk if k.split("<")[0] in ("k_bounds", "k_bounds_final", "k_morton", "k_onesweep", "k_refit_local", "k_refit_mid", "k_refit_spans")]
per_build = {k: allk[k]["derived_mean"]["fabric_bytes"] for k in build_kernels if "fabric_bytes" in allk[k].get("derived_mean", {})}
n_tri = 1310720
ones = [k for k in per_build if k.startswith("k_onesweep")]
if len(ones) == 2:      # <true> runs once, <false> twice per build
    a = [k for k in ones if "true" in k][0]; b_ = [k for k in ones if "false" in k][0]
    x = per_build.pop(a) + 2 * per_build.pop(b_)
    per_build["k_onesweep (x3)"] = x
elif len(ones) == 1:
    per_build["k_onesweep (x3)"] = 3 * per_build.pop(ones[0])
bt = {"n_tri": n_tri, "algorithmic_bytes": 560 * n_tri, "fabric_bytes_per_build": per_build, "fabric_bytes_total": sum(per_build.values()),
      "ratio": round(sum(per_build.values()) / (560.0 * n_tri), 3),
      "note": "builds WITH surface records (texcoord / material slots of the 128-byte mesh records read, 32-byte tri_uv written): they are part of these bytes but not of the 560 B/triangle figure; like for like (no surface): build_traffic_nosurface.txt"}
json.dump(bt, open(os.path.join(dst, "build_traffic.json"), "w"), indent=1)
print("build fabric traffic / algorithmic:", bt["ratio"], per_build)
