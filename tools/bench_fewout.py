#!/usr/bin/env python3
"""AlignNet head conv (2C -> 3, + the 1x1 shortcut conv) at the four SAMM levels, batch 8: us per launch of oodgan_conv3x3_fewout2
against the two-kernel form (conv3x3_fewout + conv1x1).  OODGAN_LIB selects the library (tools/ab_build.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ood-gan-inversion_amd'))
import torch  # noqa: E402
from oodgan import samm  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for K, H in ((1024, 32), (1024, 64), (512, 128), (256, 256)):
        x = torch.randn(B, K, H, H, device=dev)
        w = torch.randn(3, K, 3, 3, device=dev) * 0.02
        w11 = torch.randn(3, K, 1, 1, device=dev) * 0.02
        sc, sh = torch.rand(B, K, device=dev) + 0.5, torch.randn(B, K, device=dev)
        sl = torch.full((3,), 0.25, device=dev)
        wt, w11t = samm.fewout_weights(w, w11)
        t2 = timeit(lambda: samm.conv3x3_fewout2(x, wt, 3, sc, sh, slope=sl, w11t=w11t, M2=3))
        t1 = timeit(lambda: (samm.conv3x3_fewout(x, w, sc, sh, slope=sl), samm.conv1x1(x, w11)))
        gb = x.numel() * 4 / 1e9
        print(f'K={K} H={H} B={B}: fewout2 {t2:.1f} us ({gb / t2 * 1e6 / 1e3:.2f} TB/s of input), fewout + conv1x1 {t1:.1f} us')


if __name__ == '__main__':
    main()
