#!/bin/bash
# Fabric traffic of the LBVH build kernels WITHOUT surface records (like for like with SURVEY §8d's 560 B per triangle):
#   scripts/build_pmc.sh <out_dir>   -> <out_dir>/pmc_summary.txt, <out_dir>/traffic.json
set -u
out="$GRAFT_REPO_ROOT/$1"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/$pass" -- python3 tools/build_only.py s1m 12 nosurface > "$out/$pass.log" 2>&1 || exit 1
done
python3 scripts/pmc_summary.py "$out" "$out/traffic.json" > "$out/pmc_summary.txt"
python3 - "$out/traffic.json" <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
n = 1310720
per = {k: v["hbm_bytes_per_launch"] * (1 if not k.startswith("k_onesweep") else (1 if "true" in k else 2)) for k, v in t.items()
       if k.split("<")[0] in ("k_bounds", "k_bounds_final", "k_morton", "k_onesweep", "k_refit_local", "k_refit_mid", "k_refit_spans")}
tot = sum(per.values())
refit = sum(v for k, v in per.items() if k.startswith("k_refit"))
print(json.dumps({"per_build": per, "total": tot, "ratio_build": round(tot / (560.0 * n), 3), "refit_bytes": refit,
                  "ratio_refit_vs_368B_per_tri": round(refit / (368.0 * n), 3)}, indent=1))
PY
