#!/usr/bin/env python3
"""CU-partitioned co-run probe: do the MFMA-bound and the HBM-bound halves of a W+ step gain from running
SIDE BY SIDE on disjoint CU sets?  Two HIP streams are created with hipExtStreamCreateWithCUMask (N CUs / the other 256 - N);
the 8-wave stride-1 conv (MFMA-bound; 155 KB of LDS, 8 waves x 256 registers: nothing co-resides with it on a CU) runs on
one, an HBM-bound producer (`to_sform`, or the blur+act+S-form producer) on the other.  Reported per N: each kernel alone on
its partition, the pair together, against the serial pair on the whole chip."""
import ctypes
import math
import os
import sys
import time

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import ops  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.init()
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(cus):
    words = [0] * 8
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
    assert rc == 0, f'hipExtStreamCreateWithCUMask rc={rc}'
    return torch.cuda.ExternalStream(st.value)


B, N = 8, 20
g = torch.Generator().manual_seed(1)
C, H = 256, 128
w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
x = torch.randn(B, C, H, H, generator=g).to(dev)
s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
xs = ops.to_sform(x, s)
wf = ops.pack_conv3x3(w, precision='f16s')
y = torch.empty(B, C, H, H, device=dev)
conv = lambda: ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=s, out=y)
res, Cm = 1024, 32
big = torch.randn(B, Cm, res, res, generator=g).to(dev)
sm = (1 + 0.3 * torch.randn(B, Cm, generator=g)).to(dev)
dst = ops.SForm(B, Cm, res, res, dev)
mem = lambda: ops.to_sform(big, sm, out=dst)
for f in (conv, mem):
    f()
torch.cuda.synchronize()


def run(pairs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st, fn in pairs:
        with torch.cuda.stream(st):
            for _ in range(N):
                fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / N


full = torch.cuda.Stream()
for _ in range(2):
    run([(full, conv)]), run([(full, mem)])
ta, tb = run([(full, conv)]), run([(full, mem)])
print(f'whole chip: conv {ta:.3f} ms, to_sform {tb:.3f} ms, serial pair {ta + tb:.3f} ms', flush=True)
for pattern in ('contiguous', 'interleaved'):
    for ncu in (64, 128, 160, 192, 224):
        if pattern == 'contiguous':
            a_set = list(range(ncu))
        else:       # spread the conv's CUs evenly over the index range (every XCD / SE keeps some of both kinds)
            a_set = sorted({int(i * 256 / ncu) for i in range(ncu)})
        b_set = [c for c in range(256) if c not in set(a_set)]
        sa, sb = masked_stream(a_set), masked_stream(b_set)
        for _ in range(2):
            run([(sa, conv), (sb, mem)])
        t_a, t_b, t_ab = run([(sa, conv)]), run([(sb, mem)]), run([(sa, conv), (sb, mem)])
        print(f'{pattern:11s} conv on {len(a_set):3d} CUs {t_a:.3f} ms | to_sform on {len(b_set):3d} CUs {t_b:.3f} ms | together {t_ab:.3f} ms '
              f'(serial whole chip {ta + tb:.3f})', flush=True)
