#!/usr/bin/env python3
"""Byte ledger of one steady-state W+ step: HBM bytes by ROLE — algorithmic (bench.wplus_step_algorithmic: every tensor a launch must
read or write, once) against measured (PMC 2*FETCH_SIZE + WRITE_SIZE of the kernels playing the role, tools/step_traffic.py), sorted by the excess.
    python tools/step_ledger.py profiles/r6_final_step_traffic_streams1.json [B] [size] [--full-records]
(--full-records: the algorithmic column for 64-byte gradient records, precision 'f16s' / the state before the hi-only records of round 6)"""
import json
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
ROLE_OF = (('upvb', 'up-conv (transposed / one-pass)'), ('t2big', 'up-conv (transposed / one-pass)'), ('t2v2', 'up-conv (transposed / one-pass)'),
           ('blur_act', 'up-conv tail (blur + noise + bias + act)'),
           ('act_bwd_blurT', 'blur^T + phase split'), ('blurT_sp', 'blur^T + phase split'),
           ('s2big', 'stride-2 conv + fused activation backward'), ('s2v2', 'stride-2 conv + fused activation backward'),
           ('s1big_kernel<true', 'conv input gradient (+ dot)'), ('stripx_kernel<true', 'conv input gradient (+ dot)'), ('strip_kernel<true', 'conv input gradient (+ dot)'),
           ('s1big_kernel<false', 'conv forward (+ ToRGB, next S-form)'), ('stripx_kernel<false', 'conv forward (+ ToRGB, next S-form)'), ('s1ring', 'conv forward (+ ToRGB, next S-form)'),
           ('torgb', 'conv forward (+ ToRGB, next S-form)'), ('rgb_finish', 'conv forward (+ ToRGB, next S-form)'), ('to_sform', 'conv forward (+ ToRGB, next S-form)'),
           ('mse', 'MSE + skip pyramid'), ('down2', 'MSE + skip pyramid'))


def main():
    import bench
    t = json.load(open(sys.argv[1]))
    pos = [v for v in sys.argv[2:] if not v.startswith('--')]
    B = int(pos[0]) if len(pos) > 0 else 8
    size = int(pos[1]) if len(pos) > 1 else 1024
    hi = '--full-records' not in sys.argv
    alg, _, roles = bench.wplus_step_algorithmic(B, size, breakdown=True, hi_records=hi)
    meas = dict.fromkeys(roles, 0.0)
    named = 0.0
    for k in t['top_kernels']:
        role = next((r for sub, r in ROLE_OF if sub in k['kernel']), None)
        if role:
            meas[role] += k['bytes_per_step']
            named += k['bytes_per_step']
    total = t['hbm_bytes_per_step']
    print(f'| role | algorithmic GB | measured GB (PMC) | excess GB | ratio |\n|---|---|---|---|---|')
    for r in sorted(roles, key=lambda r: meas[r] - roles[r], reverse=True):
        print(f'| {r} | {roles[r] / 1e9:.2f} | {meas[r] / 1e9:.2f} | {(meas[r] - roles[r]) / 1e9:+.2f} | {meas[r] / roles[r]:.2f} |')
    print(f'| kernels outside the top {len(t["top_kernels"])} of the PMC table (low-resolution layers, small kernels) | — | {(total - named) / 1e9:.2f} | | |')
    print(f'| **step** | **{alg / 1e9:.2f}** | **{total / 1e9:.2f}** | **{(total - alg) / 1e9:+.2f}** | **{total / alg:.3f}** |')


if __name__ == '__main__':
    main()
