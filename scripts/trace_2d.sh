#!/bin/bash
# Kernel trace of the 2D 1024^2 workload: per-kernel statistics, queue-gap summary and the full
# kernel timeline of one factorization, written under gpurun_out/trace_2d/.
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/trace_2d
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t2d
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/t2d -- python3 $GRAFT_REPO_ROOT/bench.py --workload sparse2d:1024 --steps 2 --warmup 1 --extras none --no-cpu-baseline > "$OUT/line.json" 2> "$OUT/err.txt"
DB=$(find /tmp/t2d -name '*_results.db' | head -1)
python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py stats "$DB" "$OUT/kernel_stats.csv"
python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py gaps "$DB" > "$OUT/queue_gaps.txt"
python3 $GRAFT_REPO_ROOT/scripts/timeline.py "$DB" 100000 > "$OUT/timeline.txt"
cat "$OUT/queue_gaps.txt"
