#!/usr/bin/env python3
"""Join the summaries of one profiling session (tools/profile_round.sh) into one row per kernel:
   kernel_table.py <kernel_stats.csv> <pmc_fetch_write.csv> <pmc_sq.csv> [--csv out.csv] [--top N]
Columns: calls and average duration in the bench run; HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KB; the
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, confirmed by the 1 GiB clone in the M2 session); clock =
GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs) / kernel duration; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel
cycles x 1024 SIMDs); LDS bank conflicts as a share of LDS-active cycles; wave states as shares of SQ_WAVE_CYCLES."""
import argparse
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:46]


def read_pmc(path):
    out = defaultdict(dict)
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            out[r['Name']][r['Counter']] = (float(r['PerDispatch']), float(r['AvgKernelNs']))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('stats')
    ap.add_argument('fetch_write')
    ap.add_argument('sq')
    ap.add_argument('--csv')
    ap.add_argument('--top', type=int, default=24)
    a = ap.parse_args()
    fw, sq = read_pmc(a.fetch_write), read_pmc(a.sq)
    rows = []
    with open(a.stats, newline='') as f:
        for r in list(csv.DictReader(f))[:a.top]:
            name = r['Name']
            c = {k: v[0] for k, v in sq.get(name, {}).items()}
            ns = {k: v[1] for k, v in sq.get(name, {}).items()}
            fetch = fw.get(name, {}).get('FETCH_SIZE', (float('nan'),))[0]
            write = fw.get(name, {}).get('WRITE_SIZE', (float('nan'),))[0]

            def pct(num, den, mul=1.0):
                return round(100.0 * c[num] / (mul * c[den]), 1) if num in c and c.get(den) else ''
            rows.append({
                'kernel': short(name), 'calls_bench': int(r['Calls']), 'share_pct': float(r['Percentage']),
                'avg_us_bench': round(float(r['AverageNs']) / 1e3, 1),
                'hbm_MB_per_launch(2*FETCH+WRITE)': round((2 * fetch + write) * 1024 / 1e6, 1),
                'fetch_MB': round(fetch * 1024 / 1e6, 1), 'write_MB': round(write * 1024 / 1e6, 1),
                'mfma_busy_pct': pct('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 1024.0 / 8.0),
                'lds_bank_conflict_pct_of_lds_active': pct('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'),
                'wait_any_pct': pct('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'),
                'wait_inst_any_pct': pct('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'),
                'active_inst_pct': pct('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'),
                'clock_GHz': round(c['GRBM_GUI_ACTIVE'] / 8.0 / ns['GRBM_GUI_ACTIVE'], 2) if 'GRBM_GUI_ACTIVE' in c else '',
            })
    w = csv.DictWriter(open(a.csv, 'w', newline='') if a.csv else sys.stdout, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)


if __name__ == '__main__':
    main()
