#!/bin/bash
# A/B library for same-box comparisons: the CURRENT sources with the listed files taken from an older commit.
#   tools/ab_build.sh <commit> <file.hip> [<file.hip> ...]   ->  ood-gan-inversion_amd/oodgan/liboodgan_hip_ab.so
# Run the workload with OODGAN_LIB=<that path> to use it (oodgan/_lib.py).
set -e
cd "$(dirname "$0")/../ood-gan-inversion_amd"
commit=$1; shift
rm -rf build_ab csrc_ab && mkdir -p csrc_ab
cp csrc/*.hip csrc/*.hpp csrc_ab/
for f in "$@"; do git show "$commit:ood-gan-inversion_amd/csrc/$f" > csrc_ab/$f; done
make CSRC=csrc_ab BUILD=build_ab LIB=oodgan/liboodgan_hip_ab.so -j8 2>&1 | grep -v "^/opt/rocm/bin/hipcc" | tail -3
ls -la oodgan/liboodgan_hip_ab.so
