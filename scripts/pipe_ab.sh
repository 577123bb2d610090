cd $GRAFT_REPO_ROOT
O=gpurun_out
run () { # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 --extras none --no-cpu-baseline > $O/ab_$tag.json 2> $O/ab_$tag.err || exit 1
  python3 -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print('$tag',round(d['ms_per_step'],1),round(d['kernel_ms_per_step'],1),d['launches_per_step'],round(d['cold_first_call_ms'],1))"
}
timeout -k 10 600 python -u -m pytest tests/test_sparse_gpu.py tests/test_edges_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/join_tests.log 2>&1 || { tail -20 $O/join_tests.log; exit 1; }
tail -2 $O/join_tests.log
run c_3d64 sparse3d:64 A=1 && run c_3d96 sparse3d:96 A=1 && run c_2d sparse2d:1024 A=1
