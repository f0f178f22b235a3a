"""Does repeated inversion leak device memory (private plan pools, launch plans, pinned buffers)?  40 inversions of the bench workload, 2 streams."""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
import torch
from oodgan import synth
from oodgan.engine import GeneratorEngine, WPlusInverter
B, size, dev = 8, 1024, torch.device('cuda:0')
eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    inv = WPlusInverter(eng)
    t0 = time.perf_counter()
    w, l = inv.invert(target, w0, noises, steps=30, streams=2 if it % 4 else 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if it % 4 == 1:
        if ref is None:
            ref = (w.clone(), l.clone())
        same = torch.equal(w, ref[0]) and torch.equal(l, ref[1])
    else:
        same = None
    free, total = torch.cuda.mem_get_info()
    print(f'inversion {it:3d}: {dt * 1e3:7.1f} ms, reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB, allocated {torch.cuda.memory_allocated() / 2**30:6.2f} GiB, '
          f'device used {(total - free) / 2**30:6.2f} GiB, bit-identical to inversion 1: {same}', flush=True)
