#!/usr/bin/env python3
"""Only the W+ loop of the bench workload (B=8, 1024², one stream, bench.py's synthetic inputs): `python tools/wplus_only.py N` runs ONE
inversion of N W+ steps and nothing else.  Profiled at two step counts under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(tools/profile_round.sh), the difference of the summed counters divided by the difference in steps is the HBM traffic of one steady-state
step — set-up (weight packing, input upload) and the first two steps (exact scales, two-pass producers) cancel (tools/step_traffic.py)."""
import os
import sys
import time

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import synth  # noqa: E402
from oodgan.engine import GeneratorEngine, WPlusInverter  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
size = 1024
dev = torch.device('cuda:0')
prec = os.environ.get('OODGAN_PRECISION', 'f16s-g2')      # bench.py's default arithmetic (round 6)
streams = int(os.environ.get('OODGAN_STREAMS', '1'))
eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size, precision=prec)
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
w, l = WPlusInverter(eng).invert(target, w0, noises, steps=steps, streams=streams)
torch.cuda.synchronize()
print(f'{steps} W+ steps ({prec}, {streams} stream(s)), batch {B}: {(time.perf_counter() - t0) * 1e3:.1f} ms, final loss {l[-1].mean().item():.6f}')
