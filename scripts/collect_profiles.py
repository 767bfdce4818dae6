#!/usr/bin/env python3
"""Copy what scripts/refresh_profiles.sh measured (gpurun_out/refresh/) into profiles/<round>/ (default r2) and rebuild profiles/traffic.json.
usage: collect_profiles.py [round-dir-name]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r2"
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
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, "pmc_summary.txt"))
with open(os.path.join(src, "gpu_tests.log")) as f:
    open(os.path.join(dst, "gpu_tests.log"), "w").write("".join(f.readlines()[-3:]))
allk = json.load(open(os.path.join(src, "traffic_all.json")))
main = [k for k in allk if k.startswith("k_render<0, false, true, 0, true")]      # WIDE64, no counters, textured, pixel kind, batch
assert len(main) == 1, list(allk)
ks_bench = json.loads(open(os.path.join(src, "ks", "bench.json")).read().strip().splitlines()[-1])
# launches of the batch kernel in one pass of `--steps 20 --warmup 5`: set-up batch (16 frames), warm-up (5), timed (10 + 10)
FRAMES_PER_LAUNCH = (32 + 5 + 20) / 3.0        # the batch kernel's launches in `bench.py --steps 20 --warmup 5`: set-up 32, warm-up 5, timed 20
t = {"workload": {"scene": "s1m", "width": 3840, "height": 2160, "mode": "shadow", "layout": "wide64", "spp": 32, "textured": True,
                  "frames_per_pipeline": 32},
     "frames_per_launch": FRAMES_PER_LAUNCH,
     "source": f"profiles/{RND}/pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE and lane passes of `python3 bench.py --no-cpu-baseline "
               "--no-fif2 --no-legs --steps 20 --warmup 5`, scripts/pmc_passes.sh; per LAUNCH of the batch kernel, a launch holding "
               f"{FRAMES_PER_LAUNCH} frames on average: set-up 32, warm-up 5, timed 20)",
     "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM/rocprofv3: gfx950 FETCH_SIZE reports half of the "
                   "bytes read; Infinity-Cache hits are counted, so this is fabric-side traffic, an upper bound on HBM bytes)",
     "k_render_fetch_size_kb_raw": allk[main[0]]["fetch_size_kb_raw"], "k_render_write_size_kb_raw": allk[main[0]]["write_size_kb_raw"],
     "k_render_hbm_bytes_per_launch": allk[main[0]]["hbm_bytes_per_launch"],
     "k_render_valu_lane_util": allk[main[0]].get("valu_lane_util"),
     "all_kernels": {k: v["hbm_bytes_per_launch"] for k, v in sorted(allk.items()) if "hbm_bytes_per_launch" in v}}
json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
b = json.loads(open(os.path.join(dst, "bench_n1.json")).read().strip().splitlines()[-1])
print("bench_n1:", b["value"], b["unit"], b["ms_per_step"], "ms; roofline", b["roofline"]["frac"], "kernel_ms", b["roofline"]["kernel_ms"],
      "traffic(old file)", b["roofline"]["traffic"], "-> new", t["k_render_hbm_bytes_per_launch"])

# build: kernel trace + stage times, and the fabric traffic of the build kernels (same PMC passes: bench.py builds 5 times)
for f in ("build_kernel_stats.csv", "build.txt"):
    shutil.copy(os.path.join(src, "build", f), os.path.join(dst, f))
for f in ("build_nosurface.txt", "build_cfg5.txt", "batch_gain.txt", "batch_gain_moving.txt", "project_scaling_batch16.txt", "project_scaling_per_frame.txt",
          "rehearsal_n2.json", "rehearsal_n4.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
build_kernels = [k for k in allk if k.split("<")[0] in ("k_bounds", "k_bounds_final", "k_morton", "k_onesweep", "k_refit_local", "k_refit_mid", "k_refit_spans")]
per_build = {k: allk[k]["hbm_bytes_per_launch"] * (3 if k.startswith("k_onesweep") else 1) for k in build_kernels}
n_tri = 1310720
bt = {"n_tri": n_tri, "algorithmic_bytes": 560 * n_tri, "fabric_bytes_per_build": per_build, "fabric_bytes_total": sum(per_build.values()),
      "ratio": round(sum(per_build.values()) / (560.0 * n_tri), 3),
      "note": "k_onesweep: 3 launches per build (both template instances averaged by launch count below); surface records (48-byte index "
              "records, 32-byte tri_uv) are part of these builds but not of the 560 B/triangle figure"}
ones = [k for k in build_kernels if k.startswith("k_onesweep")]
if len(ones) == 2:      # <true> runs once, <false> twice per build
    per_build.pop(ones[0]); per_build.pop(ones[1])
    a = [k for k in ones if "true" in k][0]; b_ = [k for k in ones if "false" in k][0]
    per_build["k_onesweep (x3)"] = allk[a]["hbm_bytes_per_launch"] + 2 * allk[b_]["hbm_bytes_per_launch"]
    bt["fabric_bytes_total"] = sum(per_build.values()); bt["ratio"] = round(bt["fabric_bytes_total"] / (560.0 * n_tri), 3)
json.dump(bt, open(os.path.join(dst, "build_traffic.json"), "w"), indent=1)
print("build fabric traffic / algorithmic:", bt["ratio"], per_build)
