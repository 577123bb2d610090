cd $GRAFT_REPO_ROOT
O=gpurun_out
run () { # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --extras none --no-cpu-baseline > $O/ab_$tag.json 2> $O/ab_$tag.err || { tail -5 $O/ab_$tag.err; exit 1; }
  python3 -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print('$tag',round(d['ms_per_step'],1),round(d['kernel_ms_per_step'],1),d['launches_per_step'],round(d['cold_first_call_ms'],1))"
}
run r0_2d sparse2d:1024 GPUQR_CU_RESERVE=0 && run r16_2d sparse2d:1024 GPUQR_CU_RESERVE=16 && run r32_2d sparse2d:1024 GPUQR_CU_RESERVE=32 && run r64_2d sparse2d:1024 GPUQR_CU_RESERVE=64 &&
run r0_64 sparse3d:64 GPUQR_CU_RESERVE=0 && run r16_64 sparse3d:64 GPUQR_CU_RESERVE=16
