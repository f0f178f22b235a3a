#!/usr/bin/env python3
"""The reference's own hot path alone — ``model(x)``: e4e encoder at 256² + OOD forward at 1024², no W+ steps
(run_ood_faceGAN_inversion.py:167-172) — for rocprofv3:

    rocprofv3 --kernel-trace -d OUT -o k -- python3 tools/forward_only.py --batch 1 --reps 20

Prints one JSON line with the host wall time per call (median) and the HIP-event split encoder / OOD forward."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ood-gan-inversion_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--streams', type=int, default=3)
    ap.add_argument('--wsteps', type=int, default=100)
    ap.add_argument('--only-full', action='store_true', help='skip the B=1 leg (kernel statistics of the full batch alone)')
    a = ap.parse_args()
    import bench
    from oodgan import synth
    dev = torch.device('cuda', 0)
    m = bench.build_full_model(a, dev)
    x = torch.cat([synth.make_images(a.size, 1, seed=1000 + g) for g in range(a.batch)]).to(dev)
    per = [synth.make_noises(a.size, 1, seed=2000 + g) for g in range(a.batch)]
    noises = [torch.cat([n[i] for n in per]).to(dev) for i in range(17)]
    r = bench.forward_only(a, m, x, noises, reps=a.reps, only_full=a.only_full)
    print(json.dumps(r[f'b{a.batch}']))


if __name__ == '__main__':
    main()
