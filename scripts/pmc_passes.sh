#!/bin/bash
# PMC counter passes for the bench (run on the GPU box through gpurun). One rocprofv3 run per pass:
# counters are not split over passes by the tool (<= 8 SQ, 4 TCC, 2 GRBM per run).
# usage: scripts/pmc_passes.sh <outdir> [bench args...]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; ctrs=$2; shift 2
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -- python3 bench.py --no-cpu-baseline --no-fif2 --no-legs "$@" > "$out/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$@" &&
run tcc "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "$@" &&
run fetch "FETCH_SIZE" "$@" &&
run write "WRITE_SIZE" "$@" &&
run lanes "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" "$@"   # lane utilisation = THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU)
