#!/bin/bash
# Profiler passes of the default bench command (BASELINE config 4, the 128^3 grid) on the GPU box: a
# kernel trace with statistics, two traffic passes (FETCH_SIZE, WRITE_SIZE) and one pass of the
# matrix-pipe counters; each its own run.  Summaries (small text files) go to gpurun_out/prof_r03/;
# the sqlite files stay on the box.  usage: profile_r03.sh kt|fw|m [workload]
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof_r03
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BENCH_METIS_INPROCESS=1
# with tens of thousands of launches queued behind each other the profiler's counter collection
# segfaults inside a launch (the same command runs through without it); the counter passes therefore
# wait for the device every 512 launches (per-launch byte counts do not depend on that)
PASS=${1:-kt}
WL=${2:-sparse3d:128}
TAG=$(echo $WL | tr ':' '_')
B="$GRAFT_REPO_ROOT/bench.py --workload $WL --steps 1 --warmup 1 --extras none --no-cpu-baseline"
S="$GRAFT_REPO_ROOT/scripts/rocpd_summary.py"
db () { find "$1" -name '*_results.db' | head -1 ; }
rm -rf /tmp/p_$PASS
case $PASS in
kt)
  timeout -s KILL 1000 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -- python3 $B > "$OUT/${TAG}_kt_line.json" 2> "$OUT/${TAG}_kt.err"
  python3 $S stats "$(db /tmp/p_kt)" "$OUT/${TAG}_kernel_stats.csv"
  python3 $S gaps "$(db /tmp/p_kt)" 3 > "$OUT/${TAG}_queue_gaps.txt" ;;
fw)
  rm -rf /tmp/p_f /tmp/p_w
  export GPUQR_SYNC_EVERY=${GPUQR_SYNC_EVERY:-512}
  timeout -s KILL 550 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -- python3 $B --no-profile > "$OUT/${TAG}_f_line.json" 2> "$OUT/${TAG}_f.err"
  date
  timeout -s KILL 550 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -- python3 $B --no-profile > "$OUT/${TAG}_w_line.json" 2> "$OUT/${TAG}_w.err"
  python3 $S pmc "$(db /tmp/p_f)" "$(db /tmp/p_w)" "$OUT/${TAG}_pmc_traffic.json" "$OUT/${TAG}_pmc_traffic.md" "$WL" \
    "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --workload $WL --steps 1 --warmup 1 --extras none --no-cpu-baseline --no-profile" ;;
m)
  export GPUQR_SYNC_EVERY=${GPUQR_SYNC_EVERY:-512}
  timeout -s KILL 1000 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES -d /tmp/p_m -- python3 $B --no-profile > "$OUT/${TAG}_m_line.json" 2> "$OUT/${TAG}_m.err"
  python3 $S counters "$(db /tmp/p_m)" "$OUT/${TAG}_pmc_mfma.md" ;;
esac
date ; echo done
