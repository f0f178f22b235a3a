#!/bin/bash
# Builds and runs the packed-fp32 reproducer (tools/probes/pkfp32_repro.hip) in both builds.
# Run from the repository root on a GPU box; output -> stdout (the committed copy: profiles/r4_pkfp32_repro.txt).
set -u
HERE=$(dirname "$0")
OUT=$HERE/build
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
[ -x $OUT/pkfp32_repro_pk ] || $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -DPK_TAG='"pk"' $HERE/pkfp32_repro.hip -o $OUT/pkfp32_repro_pk -ldl 2> $OUT/pk.log
[ -x $OUT/pkfp32_repro_nopk ] || $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -DPK_TAG='"nopk"' -Xclang -target-feature -Xclang -packed-fp32-ops $HERE/pkfp32_repro.hip -o $OUT/pkfp32_repro_nopk -ldl 2> $OUT/nopk.log
LIB=${1:-ood-gan-inversion_amd/oodgan/liboodgan_hip.so}
timeout 120 $OUT/pkfp32_repro_pk $LIB
timeout 120 $OUT/pkfp32_repro_nopk $LIB
