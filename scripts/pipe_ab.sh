cd $GRAFT_REPO_ROOT
O=gpurun_out
run () { # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 --extras none --no-cpu-baseline > $O/ab_$tag.json 2> $O/ab_$tag.err || exit 1
  python3 -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print('$tag',round(d['ms_per_step'],1),round(d['kernel_ms_per_step'],1),d['launches_per_step'],round(d['cold_first_call_ms'],1))"
}
run s0_64 sparse3d:64 GPUQR_PITCH_SKEW=0 && run s1_64 sparse3d:64 GPUQR_PITCH_SKEW=1 && run s3_64 sparse3d:64 GPUQR_PITCH_SKEW=3 && run s5_64 sparse3d:64 GPUQR_PITCH_SKEW=5 &&
run s0_96 sparse3d:96 GPUQR_PITCH_SKEW=0 && run s1_96 sparse3d:96 GPUQR_PITCH_SKEW=1 && run s3_96 sparse3d:96 GPUQR_PITCH_SKEW=3 && run s5_96 sparse3d:96 GPUQR_PITCH_SKEW=5
