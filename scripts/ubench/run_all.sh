#!/bin/bash
# Runs the micro-benchmarks that the kernel design decisions rest on; prints everything to stdout.
D=$(dirname "$0")
for cmd in "mfma_f64" "slab_copy 256 4096 128" "apply_bench 256 4096 128 5 256" "apply_bench 256 4096 256 5 256" \
           "apply64_bench 64 8192 5" "factor_bench 1 256 2048 5" "factor_bench 256 256 2048 5" "factor_bench 256 32 2048 5" "launch_gap" "graph_gap" "valu_f64" ; do
  echo "== $cmd"
  timeout -k 5 120 $D/$cmd || echo "(exit $?)"
done
