#!/bin/bash
# rocprofv3 --kernel-trace --stats of the LBVH build alone:  scripts/build_stats.sh <out_dir> [scene] [n]
set -e
out="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o t -- python3 "$GRAFT_REPO_ROOT/tools/build_only.py" "$@" > "$out/build.txt" 2> "$out/err.log"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/build_kernel_stats.csv"
cat "$out/build.txt"
python3 - "$out/build_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f'{r["Name"][:60]:60s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
