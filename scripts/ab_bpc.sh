#!/bin/bash
# Blocks per CU of ONE-frame launches (RTCORE_BLOCKS_PER_CU overrides the default) on the bench's per-frame leg.
set -e
mkdir -p gpurun_out/bpc
run() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --batch 1 --steps 20 --warmup 5 "$@" > gpurun_out/bpc/$tag.json
  python - "$tag" gpurun_out/bpc/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], d["roofline"].get("kernel_ms"), flush=True)
PY
}
for b in ${1:-default 2 3 4}; do
  if [ $b = default ]; then unset RTCORE_BLOCKS_PER_CU; else export RTCORE_BLOCKS_PER_CU=$b; fi
  run notex_$b --no-texture
  run tex_$b
  run primary_notex_$b --no-texture --mode primary
done
