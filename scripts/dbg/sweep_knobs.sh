cd $GRAFT_REPO_ROOT
export PROBE_MODES=metis-child PROBE_REPS=2
run () { echo "== $*" ; env "$@" timeout -k 10 120 python -u scripts/probe_r03.py 96x96x96 2>&1 | grep "TF/s" | cut -c100-260 ; }
run A=1
run GPUQR_WIDE_ABOVE=768
run GPUQR_WIDE_ABOVE=3072
run GPUQR_WIDE_ABOVE=1000000
run GPUQR_NARROW_BELOW=512
run GPUQR_PIPE_ROWS=64
run GPUQR_RIDE_BELOW=2048
