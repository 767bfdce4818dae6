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
for v in base packed both base packed both; do run shadow_$v $v --steps 20 --warmup 5; done
for v in base packed both; do run pt16_$v $v --mode pt --spp 16 --steps 10 --warmup 2; done
