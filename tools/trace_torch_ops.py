"""Which Python lines of the host mirror still launch torch kernels / copies inside model(x)."""
import os, sys, collections
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
import torch
import bench
from oodgan import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
a = type('A', (), dict(size=1024))()
dev = torch.device('cuda:0')
m = bench.build_full_model(a, dev)
x = torch.cat([synth.make_images(1024, 1, seed=1000 + g) for g in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(1024, 1, seed=2000 + g)[i] for g in range(B)]).to(dev) for i in range(17)]
for _ in range(3):
    m(x, noise=noises)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m(x, noise=noises)
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if not e.name.startswith('aten::'):
        continue
    dt = getattr(e, 'self_device_time_total', 0) or 0
    if dt <= 0:
        continue
    frames = [f for f in (e.stack or []) if 'oodgan' in f or 'bench.py' in f]
    where = frames[0].split('ood-gan-inversion_amd/')[-1] if frames else '(no oodgan frame)'
    cnt[(e.name, where)] += 1; tim[(e.name, where)] += dt
print(f'B={B}: aten ops with device time inside one model(x): {sum(cnt.values())} launches, {sum(tim.values()) / 1e3:.3f} ms')
for k, n in cnt.most_common(60):
    print(f'{n:4d} x {k[0]:28s} {tim[k] / 1e3:7.3f} ms  {k[1]}')
