"""Four inversions of the bench workload from a fresh process with the range guard and launch plans on: ms per step, rollbacks, plan, dispatch counters —
`python tools/hi_records_probe.py [steps] [streams]`; OODGAN_HI_RECORDS=0 switches the 32-byte hi-only gradient records off (A/B: profiles/r6_hi_records_ab.txt)."""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
import torch
from oodgan import synth, _lib
from oodgan.engine import GeneratorEngine, WPlusInverter
B, size, dev = 8, 1024, torch.device('cuda:0')
eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
STREAMS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for it in range(4):
    inv = WPlusInverter(eng)
    _lib.dispatch_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, l = inv.invert(target, w0, noises, steps=STEPS, streams=STREAMS)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / STEPS
    print(f'inversion {it}: {dt:.2f} ms/step stats {inv.last_stats} plan {inv.last_plan} loss {l[-1].mean().item():.6f} finite {bool(torch.isfinite(l).all())} s1big_xh {_lib.dispatch_count("s1big_xh")}', flush=True)
