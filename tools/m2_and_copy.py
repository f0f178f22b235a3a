#!/usr/bin/env python3
"""The M2 leg of bench.py (fp16 modulated conv, BASELINE configs[4]) and a calibration copy for the PMC traffic counters:
a 1 GiB fp32 tensor cloned 5 times (16-byte loads and stores of exactly known size)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ood-gan-inversion_amd'))
import torch  # noqa: E402
import bench  # noqa: E402

torch.cuda.set_device(0)
x = torch.empty(1 << 28, device='cuda', dtype=torch.float32).normal_()
for _ in range(5):
    y = x.clone()
torch.cuda.synchronize()
del x, y
print(json.dumps(bench.modconv_roofline(iters=10, warmup=2)))
