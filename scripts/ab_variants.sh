#!/bin/bash
# A/B of builds of the same ABI on ONE box (boxes of the pool differ by a few %): RTCORE_LIBRARY selects the build.
#   raytracer_amd/lib/variants/librtcore_<name>.so   built here beforehand (git-ignored, travels with gpurun), e.g.
#     hipcc $(HIPFLAGS) -DRT_PIXEL_NOGI_BATCH_MIN_WAVES=6 -c raytracer_amd/csrc/rt_trace.hip -o /tmp/v.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/lib/variants/librtcore_w6.so build/rt_api.o build/rt_build.o /tmp/v.o build/rt_scene.o
#   gpurun -- 'scripts/ab_variants.sh "base w6 base w6" --steps 20 --warmup 5'      ("base" = raytracer_amd/lib/librtcore.so)
# Prints, per run: variant, ms per step (wall), k_render ms per pipeline (hipEvents).  Bench lines: gpurun_out/ab_variants/.
# The records of this round's A/Bs: profiles/r3/ab_slp_vectorize.txt, ab_register_cap.txt.
set -e
names=$1; shift
mkdir -p gpurun_out/ab_variants
V=$PWD/raytracer_amd/lib/variants
i=0
for lib in $names; do
  i=$((i + 1))
  if [ "$lib" = base ]; then unset RTCORE_LIBRARY; else export RTCORE_LIBRARY=$V/librtcore_$lib.so; fi
  out=gpurun_out/ab_variants/${i}_$lib.json
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs "$@" > "$out"
  python - "$lib" "$out" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], d["roofline"].get("kernel_ms"), flush=True)
PY
done
