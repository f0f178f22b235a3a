#!/usr/bin/env python3
"""What separates the three ways of driving a W+ step — Python launch by launch, the recorded launch plan (oodgan_plan_run), hipGraph replay — on the GPU's
time line.  `python tools/gap_probe.py MODE [steps]` (MODE = python | plan | graph) runs ONE inversion of the bench workload (B=8, 1024², one stream);
under `rocprofv3 --kernel-trace` the database holds every kernel's start / end, and `python tools/gap_probe.py --db FILE` prints per kernel-to-kernel
transition: the time the device spent with NO kernel running (start[i+1] - max end so far), the summed kernel time and the span, per step of the steady state."""
import os
import sys
import time


def stats(db, skip_frac=0.3):
    import sqlite3
    c = sqlite3.connect(db)
    rows = c.execute('select start, end from kernels order by start').fetchall()
    n0 = int(len(rows) * skip_frac)            # set-up and the first steps
    rows = rows[n0:]
    busy = sum(e - s for s, e in rows)
    span = max(e for _, e in rows) - rows[0][0]
    idle, hi, gaps = 0, rows[0][1], []
    for s, e in rows[1:]:
        if s > hi:
            idle += s - hi
            gaps.append(s - hi)
        hi = max(hi, e)
    gaps.sort()
    med = gaps[len(gaps) // 2] if gaps else 0
    print(f'{os.path.basename(os.path.dirname(db))}: {len(rows)} kernels, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, device idle between kernels '
          f'{idle / 1e6:.2f} ms = {100.0 * idle / span:.2f} % of the span; {len(gaps)} gaps, median {med / 1e3:.2f} us, mean {idle / max(len(gaps), 1) / 1e3:.2f} us')


def main():
    if sys.argv[1] == '--db':
        return stats(sys.argv[2])
    mode, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ood-gan-inversion_amd'))
    from oodgan import synth
    from oodgan.engine import GeneratorEngine, WPlusInverter
    B, size, dev = 8, 1024, torch.device('cuda:0')
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
    target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
    noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
    w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
    inv = WPlusInverter(eng, use_plan=(mode == 'plan'))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w, l = inv.invert(target, w0, noises, steps=steps, streams=1, use_graph=(mode == 'graph'))
    torch.cuda.synchronize()
    print(f'{mode}: {(time.perf_counter() - t0) * 1e3 / steps:.3f} ms per step wall ({steps} steps), final loss {l[-1].mean().item():.6f}')


if __name__ == '__main__':
    main()
