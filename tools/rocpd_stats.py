#!/usr/bin/env python3
"""Kernel statistics (the `--stats` summary: calls, total / average / min / max duration, share) from a rocprofv3 rocpd
database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db on this ROCm).  With --per-grid the rows are
split by launch geometry, which separates the per-layer instances of one kernel template.

    python tools/rocpd_stats.py gpurun_out/r2_prof/x_results.db [--per-grid] [--csv out.csv]"""
import argparse
import csv
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--per-grid', action='store_true')
    ap.add_argument('--csv')
    ap.add_argument('--top', type=int, default=0)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    name = 'name' if 'name' in cols else 'kernel_name'
    key = f'{name}, grid_x, grid_y, grid_z, workgroup_x' if a.per_grid else name
    q = (f'select {key}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels '
         f'group by {key} order by sum(end - start) desc')
    rows = c.execute(q).fetchall()
    total = sum(r[-4] for r in rows)
    nk = len(rows[0]) - 5
    hdr = (['Name', 'GridX', 'GridY', 'GridZ', 'WgX'] if a.per_grid else ['Name']) + ['Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs']
    out = []
    for r in rows:
        k, (n, tot, avg, mn, mx) = list(r[:nk]), r[nk:]
        out.append(k + [n, int(tot), round(avg, 1), round(100.0 * tot / total, 2), int(mn), int(mx)])
    if a.top:
        out = out[:a.top]
    w = csv.writer(open(a.csv, 'w', newline='') if a.csv else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(hdr)
    w.writerows(out)
    if a.csv:
        print(f'{len(out)} rows, total kernel time {total / 1e6:.2f} ms -> {a.csv}')


if __name__ == '__main__':
    main()
