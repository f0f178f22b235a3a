#!/bin/bash
# Does a producer -> consumer pair at the 1024² / 512² levels hit the 256 MB Infinity Cache when the batch is small enough for its
# tensors to stay there?  Kernel statistics of the W+ loop at batch 1, 2 and 8 (one stream): compare time per image per kernel.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mall}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 1 2 8; do
  rocprofv3 --kernel-trace -d $OUT/b$b -o k -- python3 bench.py --batch $b --streams 1 --wsteps 12 --steps 1 --warmup 1 --no-cpu-baseline --no-modconv --no-single-stream --no-end-to-end --no-forward-only --no-generator-fwd --no-roofline-events > $OUT/b$b.json 2> $OUT/b$b.err
  python3 tools/rocpd_stats.py $OUT/b$b/k_results.db --per-grid --csv $OUT/b$b.csv > /dev/null
done
find $OUT -name "*.db" -delete
python3 - <<PY
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        k = r['Name'][:58]
        a = d.setdefault(k, [0, 0.0])
        a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
    return d
d = {b: load('$OUT/b%d.csv' % b) for b in (1, 2, 8)}
rows = sorted(d[8].items(), key=lambda kv: -kv[1][1])[:16]
print('kernel | total us per image: batch 1 | 2 | 8   (same number of W+ steps)')
for k, v in rows:
    line = f'{k:58s}'
    for b in (1, 2, 8):
        x = d[b].get(k)
        line += f' {x[1] / 1e3 / b:10.0f}' if x else '          -'
    print(line)
for b in (1, 2, 8):
    print('batch', b, 'kernel ms per image:', sum(v[1] for v in d[b].values()) / 1e6 / b)
PY
