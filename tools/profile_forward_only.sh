#!/bin/bash
# Kernel statistics of model(x) alone (tools/forward_only.py) at B = 8 and B = 1, per kernel and per (kernel, grid): gpurun_out/$1/
set -u
TAG=${1:-fo}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 8 1; do
  rocprofv3 --kernel-trace -d $OUT/t$b -o k -- python3 tools/forward_only.py --batch $b --reps 10 --only-full > $OUT/forward_only_b$b.json 2> $OUT/err$b.txt
  python3 tools/rocpd_stats.py $OUT/t$b/k_results.db --csv $OUT/forward_only_b${b}_kernel_stats.csv
  python3 tools/rocpd_stats.py $OUT/t$b/k_results.db --per-grid --csv $OUT/forward_only_b${b}_kernel_stats_per_grid.csv
done
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
cat $OUT/forward_only_b8.json $OUT/forward_only_b1.json
