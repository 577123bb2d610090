#!/bin/bash
# Profiler passes of the default bench command on the GPU box: a kernel trace with statistics, two
# traffic passes (FETCH_SIZE, WRITE_SIZE) and one pass of the matrix-pipe counters; each its own run.
# Summaries (small text files) go to gpurun_out/prof_r02/; the sqlite files stay on the box.
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof_r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# with 44,000 launches queued behind each other the profiler's process segfaults inside a launch (the
# same command runs through without it); waiting for the device every 4096 launches avoids that
export GPUQR_SYNC_EVERY=${GPUQR_SYNC_EVERY:-4096}
PASSES=${1:-kt,pmc}
B="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --extras none --no-cpu-baseline"
S="$GRAFT_REPO_ROOT/scripts/rocpd_summary.py"
db () { find "$1" -name '*_results.db' | head -1 ; }
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_m
if [[ $PASSES == *kt* ]]; then
echo "== kernel trace" ; date
timeout -s KILL 500 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -- python3 $B > "$OUT/kt_line.json" 2> "$OUT/kt.err"
python3 $S stats "$(db /tmp/p_kt)" "$OUT/kernel_stats.csv"
python3 $S gaps "$(db /tmp/p_kt)" > "$OUT/queue_gaps.txt"
fi
[[ $PASSES == *pmc* ]] || exit 0
echo "== FETCH_SIZE" ; date
timeout -s KILL 700 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -- python3 $B > "$OUT/f_line.json" 2> "$OUT/f.err"
echo "== WRITE_SIZE" ; date
timeout -s KILL 700 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -- python3 $B > "$OUT/w_line.json" 2> "$OUT/w.err"
python3 $S pmc "$(db /tmp/p_f)" "$(db /tmp/p_w)" "$OUT/pmc_traffic.json" "$OUT/pmc_traffic.md" "sparse3d:96" \
  "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --extras none --no-cpu-baseline"
echo "== MFMA counters" ; date
timeout -s KILL 700 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES -d /tmp/p_m -- python3 $B > "$OUT/m_line.json" 2> "$OUT/m.err"
python3 $S counters "$(db /tmp/p_m)" "$OUT/pmc_mfma.md"
date ; echo done
