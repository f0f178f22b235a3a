"""LPIPS(net='alex') on the HIP kernels — the perceptual term of the inversion loss (north_star: "W+ Adam steps against LPIPS/L2").

Host mirror of the reference's ``LPIPS_Loss`` (src/losses/lpips_loss.py:13-34: same constructor arguments, ``forward(pred, target,
normalize=True) -> (loss, None)``) over csrc/lpips.hip.  PARITY UNPINNED: the reference delegates the arithmetic to the
un-vendored, un-versioned ``lpips`` package whose pretrained weights exist neither in the reference tree nor on the build / GPU boxes;
what is built here is the published algorithm (see oracle/lpips_cpu.py), to be fed the package's state dict when one is available
(``LPIPS_Loss(state_dict=torch.load('alex.pth') | lpips.LPIPS(net='alex').state_dict())``) and seeded weights otherwise.

No autograd: weights are frozen, ``loss_and_grad`` returns the per-image value and ADDS the gradient w.r.t. ``pred`` to a caller-provided
image gradient (the MSE term's, inside the W+ loop) — forward and backward of the five taps in 22 launches, all from this library
(a recorded launch plan replays them, oodgan_plan_*).
"""
import ctypes
import math
import os

import torch

from . import _lib, ops
from ._lib import ACT_NONE, ACT_PRELU, CONV_S1, check
from .ops import _p, _stream

# AlexNet's 3x3 layers (conv3-5: 57 % of the stack's flops) on the split-f16 matrix kernels of the generator (S-form + 8-wave stride-1 kernel: fp32-class
# arithmetic at ~6x the exact-fp32 MFMA rate; ReLU = PReLU with zero slopes); 0: every layer on the exact-fp32 kernel of csrc/lpips.hip
USE_F16S_3X3 = os.environ.get('OODGAN_LPIPS_F16S', '1') != '0'

SHIFT = (-.030, -.088, -.188)          # lpips.ScalingLayer
SCALE = (.458, .448, .450)
_CONVS = (('net.slice1.0', 64, 3, 11), ('net.slice2.3', 192, 64, 5), ('net.slice3.6', 384, 192, 3), ('net.slice4.8', 256, 384, 3),
          ('net.slice5.10', 256, 256, 3))


def _pack(w, transpose_flip):
    """(Co,Ci,k,k) -> [K][k*k][round_up(M,64)] fp32 (include/oodgan.h, oodgan_conv2d_s1): forward K=Ci, M=Co; input gradient K=Co, M=Ci with
    the taps flipped.  One-time weight preparation (torch ops)."""
    Co, Ci, k, _ = w.shape
    if transpose_flip:
        w = torch.flip(w, [2, 3]).transpose(0, 1)       # (Ci,Co,k,k): output channel = ci
    M, K = w.shape[0], w.shape[1]
    Mp = (M + 63) // 64 * 64
    out = torch.zeros(K, k * k, Mp, device=w.device, dtype=torch.float32)
    out[:, :, :M] = w.reshape(M, K, k * k).permute(1, 2, 0)
    return out.contiguous()


def _conv1_as_3x3(w):
    """AlexNet conv1 (64,3,11,11), stride 4, pad 2 -> the equivalent (64,48,3,3) stride-1 VALID kernel over the 4x4 space-to-depth image
    (channel c*16 + dy*4 + dx; taps 4a+dy >= 11 are zero)."""
    Co = w.shape[0]
    w12 = torch.zeros(Co, 3, 12, 12, device=w.device, dtype=w.dtype)
    w12[:, :, :11, :11] = w
    return w12.view(Co, 3, 3, 4, 3, 4).permute(0, 1, 3, 5, 2, 4).reshape(Co, 48, 3, 3).contiguous()


class LPIPSAlex:
    """Prepared weights + the forward / backward launch sequence.  ``state``: lpips.LPIPS(net='alex') state-dict layout
    (``net.slice{i}.{j}.weight|bias``, ``lin{k}.model.1.weight``) on a ROCm device."""

    def __init__(self, state, min_max=(0.0, 1.0)):
        g = lambda k: state[k].detach().float().contiguous()
        dev = g('lin0.model.1.weight').device
        if dev.type != 'cuda':
            raise RuntimeError('LPIPSAlex needs its parameters on a ROCm device (no CPU fallback)')
        self.device = dev
        lo, hi = float(min_max[0]), float(min_max[1])
        # lpips_loss.py:27-29 maps min_max to [0,1]; lpips' normalize=True maps [0,1] to [-1,1]: v = a x + b0
        self.a, self.b0 = 2.0 / (hi - lo), -2.0 * lo / (hi - lo) - 1.0
        self.shift = (ctypes.c_float * 3)(*SHIFT)
        self.scale = (ctypes.c_float * 3)(*SCALE)
        self.layers = []
        for i, (name, co, ci, k) in enumerate(_CONVS):
            w = g(name + '.weight')
            assert tuple(w.shape) == (co, ci, k, k), (name, tuple(w.shape))
            if i == 0:
                w = _conv1_as_3x3(w)
                ks, pad = 3, 0
            else:
                ks, pad = k, (k - 1) // 2
            L = dict(wf=_pack(w, False), wb=_pack(w, True), bias=g(name + '.bias'), K=w.shape[1], M=co, ks=ks, pad=pad,
                     lin=g(f'lin{i}.model.1.weight').reshape(-1).contiguous(), f16s=None)
            if USE_F16S_3X3 and ks == 3 and pad == 1 and L['K'] % 16 == 0 and min(L['K'], co) >= 64:
                L['f16s'] = (ops.pack_conv3x3(w, precision='f16s'), ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s'),
                             torch.zeros(max(co, L['K']), device=dev))          # forward / input-gradient packings, zero PReLU slopes = ReLU
            self.layers.append(L)
        self.target = None

    # ---- launches
    def _conv(self, x, L, fwd, add=None, mask=None):
        B, K, H, W = x.shape
        if L['f16s'] is not None:
            # S-form route: power-of-two range scale measured on the tensor (exact), conversion, 8-wave split-f16 kernel
            mul2 = ops.absmax_mul2(x)
            xs = ops.to_sform(x, None, mul2, out=ops.sform_scratch(B, K, H, W, x.device, tag=7))
            if fwd:
                return ops.conv3x3(xs, L['f16s'][0], L['M'], CONV_S1, bias=L['bias'], act=ACT_PRELU, slope=L['f16s'][2][:L['M']], in_mul2=mul2)
            y = ops.conv3x3(xs, L['f16s'][1], L['K'], CONV_S1, act=ACT_NONE, in_mul2=mul2)
            if add is not None:
                check(_lib.lib().oodgan_add_mask(_p(y), _p(add), _p(mask), y.numel(), _stream()), 'add_mask')
            return y
        ks = L['ks']
        pad = L['pad'] if fwd else ks - 1 - L['pad']
        M = L['M'] if fwd else L['K']
        assert K == (L['K'] if fwd else L['M'])
        y = torch.empty(B, M, H + 2 * pad - ks + 1, W + 2 * pad - ks + 1, device=x.device, dtype=torch.float32)
        check(_lib.lib().oodgan_conv2d_s1(_p(x), _p(L['wf'] if fwd else L['wb']), _p(L['bias']) if fwd else None, _p(add), _p(mask), _p(y), B, K, M,
                                          H, W, ks, pad, 1 if fwd else 0, _stream()), 'conv2d_s1')
        return y

    def _pool(self, x, want_idx=False):
        B, C, H, W = x.shape
        y = torch.empty(B, C, (H - 3) // 2 + 1, (W - 3) // 2 + 1, device=x.device, dtype=torch.float32)
        idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8) if want_idx else None       # each window's argmax, for the backward
        check(_lib.lib().oodgan_maxpool3s2_fwd(_p(x), _p(y), _p(idx), B * C, H, W, _stream()), 'maxpool_fwd')
        return (y, idx) if want_idx else y

    def _pool_bwd(self, x, gy, add, idx=None):
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        check(_lib.lib().oodgan_maxpool3s2_bwd(_p(x), _p(gy), _p(add), _p(idx), _p(gx), B * C, H, W, _stream()), 'maxpool_bwd')
        return gx

    def taps(self, img, want_idx=False):
        """The five ReLU outputs of the AlexNet feature stack for img (B,3,H,W) in this object's min_max range (``want_idx``: + the argmax tables
        of the two max-pools, for the backward)."""
        B, _, H, W = img.shape
        x48 = torch.empty(B, 48, H // 4 + 1, W // 4 + 1, device=img.device, dtype=torch.float32)
        check(_lib.lib().oodgan_lpips_prep(_p(img), _p(x48), B, H, W, self.a, self.b0, self.shift, self.scale, _stream()), 'lpips_prep')
        t1 = self._conv(x48, self.layers[0], True)
        p1, i1 = self._pool(t1, True)
        t2 = self._conv(p1, self.layers[1], True)
        p2, i2 = self._pool(t2, True)
        t3 = self._conv(p2, self.layers[2], True)
        t4 = self._conv(t3, self.layers[3], True)
        t5 = self._conv(t4, self.layers[4], True)
        return ([t1, t2, t3, t4, t5], (i1, i2)) if want_idx else [t1, t2, t3, t4, t5]

    def _head(self, f, n1, lin, coef, mode):
        B, C, H, W = f.shape
        out = torch.empty_like(f)
        part = None
        if mode != 0:
            part = torch.empty(B, _lib.lib().oodgan_lpips_head_nparts(H * W), device=f.device, dtype=torch.float32)
        # the tap's value is the SPATIAL MEAN of d (lpips.spatial_average): its gradient carries 1 / HW
        check(_lib.lib().oodgan_lpips_head(_p(f), _p(n1), _p(lin), _p(out), _p(part), B, C, H * W, float(coef) / (H * W), mode, _stream()), 'lpips_head')
        return out, part

    def target_taps(self, target):
        """Channel-normalised taps of the target image(s) (fixed during an inversion): computed once per inversion."""
        img = target.detach().float().contiguous()
        return [self._head(f, None, None, 0.0, 0)[0] for f in self.taps(img)]

    def set_target(self, target):
        self.target = self.target_taps(target)
        return self

    def loss_and_grad(self, pred, gimg=None, grad_mul=1.0, table=None, row_dev=None, target_taps=None):
        """Per-image LPIPS(pred, target) (B,) — or, with ``table`` (nrows,B) + ``row_dev``, written to row row_dev[0] of the table (returns
        None) — and, if ``gimg`` is given, gimg += grad_mul * d(sum_b lpips_b)/d(pred).  The target: ``target_taps`` (from
        ``target_taps(target)``; the W+ loop's sub-batches each keep their own) or the one of ``set_target``."""
        tgt = self.target if target_taps is None else target_taps
        assert tgt is not None and tgt[0].shape[0] == pred.shape[0]
        img = pred.detach().float().contiguous()
        B, _, H, W = img.shape
        f, (pidx1, pidx2) = self.taps(img, want_idx=True)
        hg, parts = [], []
        for k in range(5):
            g_, part = self._head(f[k], tgt[k], self.layers[k]['lin'], grad_mul, 2 if k == 4 else 1)     # the deepest tap masks itself
            hg.append(g_)
            parts.append(part)
        L = _lib.lib()
        n = 5
        parr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in parts])
        narr = (ctypes.c_int * n)(*[p.shape[1] for p in parts])
        harr = (ctypes.c_long * n)(*[t.shape[2] * t.shape[3] for t in f])
        if table is None:
            out = torch.empty(B, device=img.device, dtype=torch.float32)
            check(L.oodgan_lpips_finish(parr, narr, harr, n, _p(out), None, 1, B, _stream()), 'lpips_finish')
        else:
            out = None
            check(L.oodgan_lpips_finish(parr, narr, harr, n, _p(table), _p(row_dev), table.shape[0], B, _stream()), 'lpips_finish')
        if gimg is not None:
            assert gimg.shape == img.shape and gimg.is_contiguous() and gimg.dtype == torch.float32
            # backward through the stack: at every tap the head's gradient joins the back-propagated one, then the ReLU mask of that tap
            g4 = self._conv(hg[4], self.layers[4], False, add=hg[3], mask=f[3])
            g3 = self._conv(g4, self.layers[3], False, add=hg[2], mask=f[2])
            gp2 = self._conv(g3, self.layers[2], False)                        # gradient w.r.t. pool2's output
            g2 = self._pool_bwd(f[1], gp2, hg[1], pidx2)
            gp1 = self._conv(g2, self.layers[1], False)
            g1 = self._pool_bwd(f[0], gp1, hg[0], pidx1)
            g48 = self._conv(g1, self.layers[0], False)                        # 3x3 "full": (B,48,H/4+1,W/4+1)
            check(L.oodgan_lpips_img_grad(_p(g48), _p(gimg), B, H, W, self.a, 1.0, self.scale, _stream()), 'lpips_img_grad')
        return out


class LPIPS_Loss(torch.nn.Module):
    """Drop-in for the reference's ``LPIPS_Loss`` (src/losses/lpips_loss.py:13-34): ``forward(pred, target, normalize=True) -> (l, None)``
    with l = reduce(lpips(pred01, target01, normalize=True)) * loss_weight.  ``model_path`` / ``state_dict``: the lpips package's weights;
    without either the seeded stand-in (``synth.lpips_state``) is used and a warning is attached to the instance (``self.seeded``)."""

    def __init__(self, loss_weight=1.0, min_max=(0, 1), net='alex', model_path=None, reduction='mean', device='hip', state_dict=None):
        super().__init__()
        if net != 'alex':
            raise NotImplementedError("LPIPS_Loss: only net='alex' (the reference's default and the YAMLs' choice) is built")
        if reduction not in ('none', 'mean', 'sum'):
            raise ValueError(f'Unsupported reduction mode: {reduction}')
        self.loss_weight, self.min_max, self.reduction = loss_weight, tuple(min_max), reduction
        if state_dict is None and model_path is not None:
            state_dict = torch.load(model_path, map_location='cpu')
        self.seeded = state_dict is None
        if state_dict is None:
            from .synth import lpips_state
            state_dict = lpips_state(0)
        self._state = {k: v for k, v in state_dict.items()}
        self._net = None

    def _build(self, device):
        if self._net is None or self._net.device != device:
            self._net = LPIPSAlex({k: v.to(device) for k, v in self._state.items()}, self.min_max)
        return self._net

    def forward(self, pred, target, normalize=True):
        if not normalize:
            raise NotImplementedError('LPIPS_Loss: the reference always calls with normalize=True (lpips_loss.py:24-31)')
        net = self._build(pred.device)
        net.set_target(target)
        per = net.loss_and_grad(pred)
        if self.reduction == 'mean':
            per = per.mean()
        elif self.reduction == 'sum':
            per = per.sum()
        return per * self.loss_weight, None
