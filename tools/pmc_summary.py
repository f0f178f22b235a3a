#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd database (one row per kernel name and counter):
    python tools/pmc_summary.py DIR_OR_DB [...] [--csv out.csv]
SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md); SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over SIMDs."""
import argparse
import csv
import glob
import os
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('paths', nargs='+')
    ap.add_argument('--csv')
    a = ap.parse_args()
    rows = {}
    for path in a.paths:
        dbs = [path] if path.endswith('.db') else glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
        for db in dbs:
            c = sqlite3.connect(db)
            q = ('select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection '
                 'group by kernel_name, counter_name')
            try:
                res = c.execute(q).fetchall()
            except sqlite3.OperationalError:
                cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
                print('counters_collection columns:', cols, file=sys.stderr)
                raise
            for name, ctr, n, tot, avg_ns in res:
                rows[(name, ctr)] = (n, tot, avg_ns)
    out = [['Name', 'Counter', 'Dispatches', 'Sum', 'PerDispatch', 'AvgKernelNs']]
    for (name, ctr), (n, tot, avg_ns) in sorted(rows.items()):
        out.append([name, ctr, n, tot, tot / n, round(avg_ns, 1)])
    w = csv.writer(open(a.csv, 'w', newline='') if a.csv else sys.stdout)
    w.writerows(out)


if __name__ == '__main__':
    main()
