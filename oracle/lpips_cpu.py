"""CPU ORACLE of the LPIPS(net='alex') loss term — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ref_cpu.py for the rules).

PARITY UNPINNED.  The reference calls the third-party ``lpips`` package (src/losses/lpips_loss.py:14,17,31; src/metrics/lpips.py:29,74);
the package is not in /root/reference, no version is pinned anywhere in it, it is not installed here and its pretrained
weights (AlexNet features + the five `lin` layers) are absent.  This file restates the PUBLISHED algorithm — Zhang et al., "The
Unreasonable Effectiveness of Deep Features as a Perceptual Metric" (CVPR 2018), as implemented by lpips 0.1.x (``lpips/lpips.py``:
``LPIPS.forward``, ``ScalingLayer``, ``NetLinLayer``, ``normalize_tensor``, ``spatial_average``; ``lpips/pretrained_networks.py``:
``alexnet`` with torchvision's AlexNet ``features`` stack sliced after each of the five ReLUs) — in plain torch, on SEEDED weights
(``oodgan.synth.lpips_state``).  What IS anchored on the reference: the wrapper's range handling and reduction
(src/losses/lpips_loss.py:24-34: map min_max to [0,1], call with normalize=True, reduce 'mean', times loss_weight).
No golden vector can exist for it; tests compare the HIP ops with this restatement and its autograd only.
"""
import torch
import torch.nn.functional as F

SHIFT = (-.030, -.088, -.188)          # lpips.ScalingLayer
SCALE = (.458, .448, .450)
CHANNELS = (64, 192, 384, 256, 256)    # AlexNet relu1 .. relu5


def alexnet_taps(P, x):
    """torchvision.models.alexnet().features, taps after each ReLU (lpips/pretrained_networks.py: alexnet.forward)."""
    h = F.relu(F.conv2d(x, P['net.slice1.0.weight'], P['net.slice1.0.bias'], stride=4, padding=2))
    t1 = h
    h = F.max_pool2d(h, kernel_size=3, stride=2)
    h = F.relu(F.conv2d(h, P['net.slice2.3.weight'], P['net.slice2.3.bias'], padding=2))
    t2 = h
    h = F.max_pool2d(h, kernel_size=3, stride=2)
    h = F.relu(F.conv2d(h, P['net.slice3.6.weight'], P['net.slice3.6.bias'], padding=1))
    t3 = h
    h = F.relu(F.conv2d(h, P['net.slice4.8.weight'], P['net.slice4.8.bias'], padding=1))
    t4 = h
    h = F.relu(F.conv2d(h, P['net.slice5.10.weight'], P['net.slice5.10.bias'], padding=1))
    return [t1, t2, t3, t4, h]


def normalize_tensor(f, eps=1e-10):
    """lpips.normalize_tensor"""
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + eps)


def lpips_alex(P, in0, in1, normalize=True):
    """lpips.LPIPS(net='alex', lpips=True, spatial=False).forward(in0, in1, normalize) -> (B,1,1,1)"""
    if normalize:                       # [0,1] -> [-1,1]
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    shift = torch.tensor(SHIFT, dtype=in0.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=in0.dtype).view(1, 3, 1, 1)
    f0 = alexnet_taps(P, (in0 - shift) / scale)
    f1 = alexnet_taps(P, (in1 - shift) / scale)
    val = 0
    for k in range(5):
        d = (normalize_tensor(f0[k]) - normalize_tensor(f1[k])) ** 2
        val = val + F.conv2d(d, P[f'lin{k}.model.1.weight']).mean(dim=(2, 3), keepdim=True)
    return val


def lpips_loss(P, pred, target, loss_weight=1.0, min_max=(0.0, 1.0), reduction='mean'):
    """LPIPS_Loss.forward (src/losses/lpips_loss.py:24-34) -> (loss, per-image values (B,))."""
    lo, hi = min_max
    pred = (pred - lo) / (hi - lo)
    target = (target - lo) / (hi - lo)
    per = lpips_alex(P, pred, target, normalize=True).flatten()
    red = per.mean() if reduction == 'mean' else (per.sum() if reduction == 'sum' else per)
    return red * loss_weight, per
