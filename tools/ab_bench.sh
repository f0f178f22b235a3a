#!/bin/bash
# Same-box A/B of two builds of the library on the bench workload (one HIP stream, 40 W+ steps): alternates the product library
# and OODGAN_LIB=<variant>, three rounds.  Usage (on the GPU box): tools/ab_bench.sh ood-gan-inversion_amd/oodgan/liboodgan_hip_ab.so
V=$(readlink -f "$1")
ARGS="--streams 1 --wsteps 40 --steps 1 --warmup 1 --no-cpu-baseline --no-modconv --no-single-stream --no-end-to-end --no-forward-only --no-generator-fwd --no-roofline-events --no-b1 --no-lpips --no-precision-ab"
for r in 1 2 3; do
  for lib in product variant; do
    if [ $lib = variant ]; then export OODGAN_LIB=$V; else unset OODGAN_LIB; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'round $r', 'ms per inversion of 8 (40 steps + OOD forward):', d['ms_per_step'])"
  done
done
