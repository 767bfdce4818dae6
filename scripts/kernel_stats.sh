#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's bench command (--steps 20 --warmup 5; run on the GPU box through gpurun):
#   scripts/kernel_stats.sh <out_dir> [bench args...]   -> <out_dir>/kernel_stats.csv, <out_dir>/bench.json
set -e
out="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o t -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-legs --steps 20 --warmup 5 "$@" > "$out/bench.json" 2> "$out/err.log"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
cut -c1-260 "$out/bench.json"
