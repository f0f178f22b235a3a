#!/bin/bash
# Same-box per-kernel A/B: kernel-trace of the shortened bench workload (one stream, 20 W+ steps) under the product library and
# under OODGAN_LIB=<variant>, per-grid statistics of both side by side.  Usage (on the GPU box): tools/ab_profile.sh <variant.so> [tag]
set -u
V=$(readlink -f "$1")
TAG=${2:-abprof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--streams 1 --wsteps 20 --steps 1 --warmup 1 --no-cpu-baseline --no-modconv --no-single-stream --no-end-to-end --no-forward-only --no-generator-fwd --no-roofline-events"
for lib in product variant; do
  if [ $lib = variant ]; then export OODGAN_LIB=$V; else unset OODGAN_LIB; fi
  rocprofv3 --kernel-trace -d $OUT/$lib -o k -- python3 bench.py $ARGS > $OUT/$lib.json 2> $OUT/$lib.err
  python3 tools/rocpd_stats.py $OUT/$lib/k_results.db --per-grid --csv $OUT/$lib.csv > /dev/null
done
find $OUT -name "*.db" -delete
python3 - <<PY
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[(r['Name'][:60], r['GridX'], r['GridY'])] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
    return d
a, b = load('$OUT/product.csv'), load('$OUT/variant.csv')
rows = []
for k in a:
    if k in b and a[k][0] >= 20:
        rows.append((a[k][0] * (a[k][1] - b[k][1]), k, a[k], b[k]))
rows.sort()
print('kernel | grid | calls | product us | variant us | total diff ms')
for d, k, x, y in rows[:25] + rows[-10:]:
    print(f'{k[0]:60s} {k[1]:>8s} {x[0]:5d} {x[1]:9.1f} {y[1]:9.1f} {d / 1e3:8.2f}')
print('sum', sum(r[0] for r in rows) / 1e3, 'ms over the run')
PY
