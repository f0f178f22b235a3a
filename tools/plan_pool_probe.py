"""Launch-plan allocator pools (oodgan/engine.py: _plan_pool): inversions of B=8 (two streams, one stream) and B=1 alternate on one process; per inversion the wall
time, the plan and the reserved memory.  With ONE cached pool per stream a B=1 step recorded into blocks a B=8 step left behind ran 2.6x slower; pools are cached per
(stream, batch size).  OODGAN_PLAN_POOL_CACHE=0: a fresh pool per inversion."""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file lives in tools/)
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
import torch
from oodgan import synth
from oodgan.engine import GeneratorEngine, WPlusInverter
size, dev = 1024, torch.device('cuda:0')
eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
def data(B):
    target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
    noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
    w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
    return target, w0, noises
d8, d1 = data(8), data(1)
def run(tag, d, streams):
    inv = WPlusInverter(eng)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inv.invert(*d, steps=100, streams=streams)
    torch.cuda.synchronize()
    print(f'{tag}: {(time.perf_counter() - t0) * 1e3:.1f} ms, plan {inv.last_plan}, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB', flush=True)
for rnd in range(2):
    run('B=8 two streams', d8, 2)
    run('B=8 one stream', d8, 1)
    run('B=1', d1, 1)
    run('B=1 again', d1, 1)
