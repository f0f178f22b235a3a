"""LPIPS(alex) forward + backward to the image alone (B=8, 1024², seeded weights): ms per call; under rocprofv3 --kernel-trace the per-kernel split
(profiles/r6_lpips_only_kernel_stats_per_grid.csv)."""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
import torch
from oodgan import synth
from oodgan.lpips import LPIPSAlex
B, size, dev = 8, 1024, torch.device('cuda:0')
net = LPIPSAlex({k: v.to(dev) for k, v in synth.lpips_state(0).items()}, min_max=(-1.0, 1.0))
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
pred = torch.cat([synth.make_images(size, 1, seed=5000 + i) for i in range(B)]).to(dev)
net.set_target(target)
gimg = torch.zeros_like(pred)
for _ in range(3):
    net.loss_and_grad(pred, gimg, 4.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    net.loss_and_grad(pred, gimg, 4.0)
torch.cuda.synchronize()
print(f'LPIPS forward + backward to the image, B={B} {size}²: {(time.perf_counter() - t0) * 1e3 / 20:.3f} ms per call')
