#!/bin/bash
# PMC counter passes for the bench (run on the GPU box through gpurun). One rocprofv3 run per pass:
# counters are not split over passes by the tool (<= 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 GRBM per run; FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
# usage: scripts/pmc_passes.sh <outdir> [bench args...]        (CALIB=1: the same passes over tools/pmc_calib instead of bench.py)
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; ctrs=$2; shift 2
  if [ "${CALIB:-0}" = 1 ]; then
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -- "$GRAFT_REPO_ROOT/tools/pmc_calib" > "$out/$name.log" 2>&1
  else
    timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -- python3 bench.py --no-cpu-baseline --no-legs "$@" > "$out/$name.log" 2>&1
  fi
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$@" &&
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE" "$@" &&
run tcc2 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_SECTORS_sum TCC_TAG_STALL_sum" "$@" &&
run tcp "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "$@" &&
run fetch "FETCH_SIZE" "$@" &&
run write "WRITE_SIZE" "$@" &&
run lanes "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" "$@"   # lane utilisation = THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU)
