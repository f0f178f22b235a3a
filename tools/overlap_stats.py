#!/usr/bin/env python3
"""Concurrency of a multi-stream run from a rocprofv3 kernel trace (rocpd database): how much of the time 0 / 1 / 2 / ... kernels are on the
GPU at once, and for every pair of kernel families the time they spent side by side.
    python tools/overlap_stats.py gpurun_out/x/k_results.db [--from-frac 0.3] [--top 25]
Only the window [--from-frac, 1] of the trace is analysed (the steady state of the last timed inversion)."""
import argparse
import sqlite3
from collections import defaultdict


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--from-frac', type=float, default=0.5)
    ap.add_argument('--to-frac', type=float, default=0.95)
    ap.add_argument('--top', type=int, default=25)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    name = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute(f'select {name}, start, end from kernels order by start').fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo, hi = t0 + a.from_frac * (t1 - t0), t0 + a.to_frac * (t1 - t0)
    ev = []
    for n, s, e in rows:
        s, e = max(s, lo), min(e, hi)
        if e > s:
            ev.append((s, 1, short(n)))
            ev.append((e, -1, short(n)))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = defaultdict(int)
    depth_time = defaultdict(float)
    pair_time = defaultdict(float)
    solo_time = defaultdict(float)
    busy = defaultdict(float)
    prev = lo
    for t, d, n in ev:
        dt = t - prev
        if dt > 0:
            names = [k for k, v in active.items() for _ in range(v)]
            depth_time[len(names)] += dt
            for k in names:
                busy[k] += dt
            if len(names) == 1:
                solo_time[names[0]] += dt
            elif len(names) >= 2:
                for i in range(len(names)):
                    for j in range(i + 1, len(names)):
                        pair_time[tuple(sorted((names[i], names[j])))] += dt
        active[n] += d
        if active[n] == 0:
            del active[n]
        prev = t
    span = hi - lo
    print(f'window {span / 1e6:.2f} ms')
    for k in sorted(depth_time):
        print(f'  {k} kernels on the GPU: {100 * depth_time[k] / span:5.1f} %')
    print('kernel family: busy ms, alone %')
    for k, v in sorted(busy.items(), key=lambda x: -x[1])[:a.top]:
        print(f'  {k:46s} {v / 1e6:8.2f} ms  alone {100 * solo_time[k] / v:5.1f} %')
    print('pairs side by side (ms):')
    for (x, y), v in sorted(pair_time.items(), key=lambda x: -x[1])[:a.top]:
        print(f'  {x:44s} | {y:44s} {v / 1e6:8.2f}')


if __name__ == '__main__':
    main()
