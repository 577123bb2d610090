cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace -d /tmp/kt2d -- python3 $R/bench.py --workload sparse2d:1024 --steps 2 --warmup 1 --extras none --no-cpu-baseline > /tmp/o 2>/tmp/e
DB=$(find /tmp/kt2d -name '*_results.db' | head -1)
python3 $R/scripts/timeline.py $DB 130 0 > $R/gpurun_out/tl_2d_tail.txt
python3 $R/scripts/timeline.py $DB 130 1500 > $R/gpurun_out/tl_2d_mid.txt
python3 $R/scripts/rocpd_summary.py gaps $DB > $R/gpurun_out/gaps_2d.txt
python3 $R/scripts/rocpd_summary.py stats $DB $R/gpurun_out/stats_2d.csv
