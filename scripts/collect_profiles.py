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
main = [k for k in allk if k.startswith("k_render<0, false, true, 0")]
assert len(main) == 1, list(allk)
t = {"workload": {"scene": "s1m", "width": 3840, "height": 2160, "mode": "shadow", "layout": "wide64", "spp": 32, "textured": True},
     "source": f"profiles/{RND}/pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE and lane passes of `python3 bench.py --no-cpu-baseline "
               "--no-fif2 --no-legs --steps 5 --warmup 3`, scripts/pmc_passes.sh)",
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
