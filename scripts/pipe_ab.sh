# A/B runs of schedule knobs on the 2D workload (and 3D 64^3): prints step / kernel ms per setting.
cd $GRAFT_REPO_ROOT
O=gpurun_out
run () { # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --extras none --no-cpu-baseline > $O/ab_$tag.json 2> $O/ab_$tag.err || exit 1
  python3 -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print('$tag',round(d['ms_per_step'],1),round(d['kernel_ms_per_step'],1),d['launches_per_step'])"
}
run p32 sparse2d:1024 GPUQR_PIPE_ROWS=32 &&
run p32_prio sparse2d:1024 GPUQR_PIPE_ROWS=32 GPUQR_STREAM_PRIORITY=1 &&
run 3d_p32 sparse3d:64 GPUQR_PIPE_ROWS=32 &&
run 3d_p32_prio sparse3d:64 GPUQR_PIPE_ROWS=32 GPUQR_STREAM_PRIORITY=1
