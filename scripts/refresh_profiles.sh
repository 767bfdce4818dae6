#!/bin/bash
# Everything profiles/<round> holds, measured in ONE gpurun call (one box: boxes differ by a few % in k_render time).
#   gpurun --timeout 1200 -- 'scripts/refresh_profiles.sh'   then, here:   python scripts/collect_profiles.py
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/refresh"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
echo "[1] gpu tests"; timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > "$O/gpu_tests.log" 2>&1; tail -1 "$O/gpu_tests.log"
echo "[2] the driver's bench line"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_n1.json" 2> "$O/bench_n1.err"
echo "[3] kernel trace"; scripts/kernel_stats.sh gpurun_out/refresh/ks > "$O/ks.txt" 2>&1; cd "$R"
echo "[4] pmc passes (calibration kernels, then the bench)"
CALIB=1 scripts/pmc_passes.sh "$O/calib" > "$O/calib.txt" 2>&1; cd "$R"
python3 scripts/pmc_summary.py "$O/calib" > "$O/pmc_calib.txt"; rm -rf "$O"/calib/*/
scripts/pmc_passes.sh "$O/pmc" --steps 20 --warmup 5 > "$O/pmc.txt" 2>&1; cd "$R"
python3 scripts/pmc_summary.py "$O/pmc" "$O/traffic_all.json" > "$O/pmc_summary.txt"; rm -rf "$O"/pmc/*/
echo "[5] variants"
b() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > "$O/bench_$name.json" 2> "$O/bench_$name.err"; echo "  $name done"; }
b notexture --no-texture --no-legs
b primary --mode primary --no-legs
b aos32 --layout aos32 --no-legs
b pt16 --mode pt --spp 16 --steps 10 --no-legs
b s70k_1080p --scene s70k --width 1920 --height 1080 --mode primary
RTCORE_TWO_PHASE=1 b traverse_shade_split --no-legs
echo "[6] cfg 5"; timeout -k 10 500 python bench.py --no-cpu-baseline --no-legs --scene lattice76 --width 7680 --height 4320 --steps 10 > "$O/bench_cfg5_100Mtri_8k.json" 2> "$O/bench_cfg5.err"
echo "[7] frag"; b frag_root --mode frag-root --steps 10 --no-legs; b frag_web --mode frag-web --steps 5 --no-legs
echo "[8] build kernels"; scripts/build_stats.sh gpurun_out/refresh/build s1m 20 > "$O/build_stats.txt" 2>&1; cd "$R"
timeout -k 10 120 python -u tools/build_only.py s1m 30 nosurface > "$O/build_nosurface.txt" 2>&1
timeout -k 10 300 python -u tools/build_only.py lattice76 6 nosurface > "$O/build_cfg5.txt" 2>&1
scripts/build_pmc.sh gpurun_out/refresh/build_pmc > "$O/build_traffic_nosurface.txt" 2>&1; cd "$R"; rm -rf "$O"/build_pmc/*/
echo "[9] N > 1 rehearsal on this one GPU (gloo through host memory: code path only, not a measurement)"
scripts/rehearse_multi.sh 2 4 > "$O/rehearsal.txt" 2>&1 || true
for n in 2 4; do [ -f "$R/gpurun_out/rehearse/bench_n$n.json" ] && tail -1 "$R/gpurun_out/rehearse/bench_n$n.json" > "$O/rehearsal_n$n.json"; done
echo "[10] per-rank projection of the N > 1 loop on this one GPU (tools/project_scaling.py)"
PROJ_BATCH=16 timeout -k 10 300 python tools/project_scaling.py 2>&1 | grep "^N=" > "$O/project_scaling_batch16.txt" || true
RTCORE_TWO_PHASE=1 PROJ_BATCH=16 timeout -k 10 300 python tools/project_scaling.py 2>&1 | grep "^N=" > "$O/project_scaling_batch16_split.txt" || true
PROJ_BATCH=1 timeout -k 10 300 python tools/project_scaling.py 2>&1 | grep "^N=" > "$O/project_scaling_per_frame.txt" || true
echo done
