#!/usr/bin/env python3
"""HBM traffic of one steady-state W+ step from PMC summaries (tools/pmc_summary.py CSVs) of `tools/wplus_only.py` at two step counts:
    python tools/step_traffic.py pmc_n1.csv N1 pmc_n2.csv N2 [--json-key wplus_step_f16s_b8_s1024]
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 summed over every dispatch of the process (the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte requests as 32); per step = (total(N2) - total(N1)) / (N2 - N1)."""
import csv
import json
import sys


def total(path):
    f = w = 0.0
    per = {}
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            v = float(r['Sum'])
            if r['Counter'] == 'FETCH_SIZE':
                f += v
                per.setdefault(r['Name'], [0.0, 0.0])[0] += v
            elif r['Counter'] == 'WRITE_SIZE':
                w += v
                per.setdefault(r['Name'], [0.0, 0.0])[1] += v
    return (2 * f + w) * 1024.0, per


def main():
    p1, n1, p2, n2 = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    t1, per1 = total(p1)
    t2, per2 = total(p2)
    step = (t2 - t1) / (n2 - n1)
    rows = []
    for k, (f2, w2) in per2.items():
        f1, w1 = per1.get(k, (0.0, 0.0))
        rows.append(((2 * (f2 - f1) + (w2 - w1)) * 1024.0 / (n2 - n1), k))
    rows.sort(reverse=True)
    out = dict(hbm_bytes_per_step=step, steps=[n1, n2], top_kernels=[dict(kernel=k.replace('(anonymous namespace)::', '')[:70], bytes_per_step=round(b)) for b, k in rows[:14]])
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
