# A/B of builds of the same ABI on one box (RTCORE_LIBRARY selects the build): bench lines into gpurun_out/ab_variants/
set -e
mkdir -p gpurun_out/ab_variants
V=$PWD/raytracer_amd/lib/variants
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], d["roofline"].get("kernel_ms"), flush=True)
PY
}
run() { tag=$1; lib=$2; shift 2
  if [ $lib = base ]; then unset RTCORE_LIBRARY; else export RTCORE_LIBRARY=$V/librtcore_$lib.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs "$@" > gpurun_out/ab_variants/$tag.json
  line $tag gpurun_out/ab_variants/$tag.json; }
run shadow_base_1 base --steps 20 --warmup 5
run shadow_nogi_w4 nogi_w4 --steps 20 --warmup 5
run shadow_base_2 base --steps 20 --warmup 5
RTCORE_TWO_PHASE=1 run shadow_split base --steps 20 --warmup 5
run pt16_base base --mode pt --spp 16 --steps 10 --warmup 2
run pt16_w4 pt_w4 --mode pt --spp 16 --steps 10 --warmup 2
run pt16_w6 pt_w6 --mode pt --spp 16 --steps 10 --warmup 2
run fragweb_base base --mode frag-web --steps 5 --warmup 1
run per_frame base --batch 1 --steps 20 --warmup 5
unset RTCORE_LIBRARY
timeout -k 10 120 python tools/build_only.py s1m 20 > gpurun_out/ab_variants/build_noslp.txt && tail -3 gpurun_out/ab_variants/build_noslp.txt
RTCORE_LIBRARY=$V/librtcore_buildslp.so timeout -k 10 120 python tools/build_only.py s1m 20 > gpurun_out/ab_variants/build_slp.txt && tail -3 gpurun_out/ab_variants/build_slp.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/ab_variants/gpu_tests.log 2>&1; tail -3 gpurun_out/ab_variants/gpu_tests.log
