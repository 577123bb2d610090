# A/B runs of schedule knobs: prints step / kernel ms per setting.
cd $GRAFT_REPO_ROOT
O=gpurun_out
run () { # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 2 --extras none --no-cpu-baseline > $O/ab_$tag.json 2> $O/ab_$tag.err || exit 1
  python3 -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print('$tag',round(d['ms_per_step'],1),round(d['kernel_ms_per_step'],1),d['launches_per_step'])"
}
for wl in sparse3d:64 sparse2d:1024; do
run base_$wl $wl A=1 &&
run wide768_$wl $wl GPUQR_WIDE_ABOVE=768 &&
run wide3072_$wl $wl GPUQR_WIDE_ABOVE=3072 &&
run wide0_$wl $wl GPUQR_WIDE_ABOVE=0 &&
run narrow64_$wl $wl GPUQR_NARROW_BELOW=64 &&
run narrow256_$wl $wl GPUQR_NARROW_BELOW=256 &&
run narrow512_$wl $wl GPUQR_NARROW_BELOW=512 &&
run ride0_$wl $wl GPUQR_RIDE_BELOW=0 &&
run ride1024_$wl $wl GPUQR_RIDE_BELOW=1024 &&
run ride2048_$wl $wl GPUQR_RIDE_BELOW=2048 &&
run small1_$wl $wl GPUQR_SMALL_ON=1 &&
run small0_$wl $wl GPUQR_SMALL_ON=0 || exit 1
done
