#!/bin/bash
# Occupancy / wait / texture-path counters of the LBVH build kernels (nosurface S1M builds), two rocprofv3 passes:
#   scripts/build_pmc_sq.sh <out_dir>   -> <out_dir>/pmc_summary.txt
set -u
out="$GRAFT_REPO_ROOT/$1"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$out/sq" -- python3 tools/build_only.py s1m 8 nosurface > "$out/sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d "$out/tcp" -- python3 tools/build_only.py s1m 8 nosurface > "$out/tcp.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE --output-format csv -d "$out/tcc" -- python3 tools/build_only.py s1m 8 nosurface > "$out/tcc.log" 2>&1 || exit 1
python3 scripts/pmc_summary.py "$out" "$out/traffic.json" > "$out/pmc_summary.txt"
