"""Roofline run of the fp16 modulated conv (BASELINE.json configs[4]): x (16,32,1024,1024) f16, 32 -> 32 channels.
Algorithmic bytes: 2*(B*Ci*H^2 + B*Co*H^2 + Co*Ci*9) + 2*B*(512 + Ci) = 2.147 GB; 309 GFLOP."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ood-gan-inversion_amd'))


def run(B=16, C=32, H=1024, iters=50, warmup=5):
    from oodgan import ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    xh = ops.HForm(B, C, H, H, dev)
    for b in range(B):                      # fill sample by sample: the fp32 staging tensor stays small
        ops.to_hform(torch.randn(1, C, H, H, generator=g).to(dev), out=_view(xh, b))
    wgt = torch.randn(1, C, C, 3, 3, generator=g).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    noise = torch.randn(B, 1, H, H, generator=g).to(dev)
    bias = (0.1 * torch.randn(C, generator=g)).to(dev)
    nw = torch.tensor([0.1], device=dev)
    packed = ops.modconv_f16_pack(wgt, s, act='lrelu')
    out = ops.HForm(B, C, H, H, dev)
    for _ in range(warmup):
        ops.modconv_f16(xh, packed, noise, nw, bias, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.modconv_f16(xh, packed, noise, nw, bias, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    alg = 2 * (B * C * H * H + B * C * H * H + C * C * 9) + 2 * B * (512 + C)
    flops = 2.0 * B * C * C * 9 * H * H
    return {'kernel': 'modconv_f16_kernel<2>', 'B': B, 'Ci': C, 'Co': C, 'H': H, 'ms': ms, 'alg_bytes': alg,
            'GBps': alg / ms / 1e6, 'hbm_frac': alg / ms / 1e6 / 8000.0, 'TFLOPs': flops / ms / 1e9}


class _view:
    """H-form view of one sample of a batch buffer (for filling)."""

    def __init__(self, h, b):
        per = h.buf.numel() // h.B
        self.buf = h.buf[b * per:(b + 1) * per]
        self.B, self.C, self.H, self.W = 1, h.C, h.H, h.W

    def data_ptr(self):
        return self.buf.data_ptr()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    print(json.dumps(run(a.batch, 32, a.size, a.iters)))
