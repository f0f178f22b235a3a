"""Image I/O and full-reference metrics — the step AFTER the accelerated path.

Same conventions as the reference harness (run_ood_faceGAN_inversion.py:64-126,159-180) without cv2:
images travel as HxWx3 **BGR** arrays (what ``cv2.imread`` returns), tensors are RGB NCHW in [-1,1].
  img2tensor / tensor2img   BasicSR/basicsr/utils/img_util.py:9-94   (clamp -> [0,1] -> x255 -> round -> uint8)
  calculate_psnr / _ssim    BasicSR/basicsr/metrics/psnr_ssim.py:9-128, metric_util.py:6-45, matlab_functions.py:214-244
  extract_masks             run_ood_faceGAN_inversion.py:74-87       (nearest up-sampling, masks side by side)
PSNR is closed form.  SSIM restates the published algorithm (11x11 Gaussian window sigma 1.5, valid region only);
cv2 is absent from this image, so SSIM is checked against a direct double-loop evaluation, not against cv2 itself.
LPIPS / identity need third-party weights that are not available: the CLI reports them as skipped."""
import math
import os

import numpy as np
import torch


def imread(path):
    """File -> HxWx3 uint8 BGR (cv2.imread convention)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return rgb[:, :, ::-1].copy()


def imwrite(path, img):
    """HxWx3 BGR (or HxW gray) uint8 -> file (cv2.imwrite convention)."""
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    arr = np.asarray(img)
    if arr.ndim == 3:
        arr = arr[:, :, ::-1]
    Image.fromarray(np.ascontiguousarray(arr)).save(path)


def img2tensor(imgs, bgr2rgb=True, float32=True):
    """HWC ndarray (or list) -> CHW tensor (or list); BGR->RGB for 3-channel input."""
    def one(img):
        if img.ndim == 3 and img.shape[2] == 3 and bgr2rgb:
            if img.dtype == np.float64:
                img = img.astype(np.float32)
            img = img[:, :, ::-1]
        t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        return t.float() if float32 else t
    return [one(i) for i in imgs] if isinstance(imgs, list) else one(imgs)


def tensor2img(tensor, rgb2bgr=True, out_type=np.uint8, min_max=(0, 1)):
    """(1x)CxHxW / HxW tensor (or list) -> HxWxC BGR / HxW ndarray; uint8 output is ROUNDED (img_util.py:87-90)."""
    if not (torch.is_tensor(tensor) or (isinstance(tensor, list) and all(torch.is_tensor(t) for t in tensor))):
        raise TypeError(f'tensor or list of tensors expected, got {type(tensor)}')
    single = torch.is_tensor(tensor)
    out = []
    for t in ([tensor] if single else tensor):
        t = t.squeeze(0).float().detach().cpu().clamp(*min_max)
        t = (t - min_max[0]) / (min_max[1] - min_max[0])
        if t.dim() == 3:
            a = t.numpy().transpose(1, 2, 0)
            if a.shape[2] == 1:
                a = a[:, :, 0]
            elif rgb2bgr:
                a = a[:, :, ::-1]
        elif t.dim() == 2:
            a = t.numpy()
        else:
            raise TypeError(f'Only 3D or 2D tensors (after squeezing the batch) are supported, got {t.dim()}D')
        if out_type == np.uint8:
            a = (a * 255.0).round()
        out.append(np.ascontiguousarray(a).astype(out_type))
    return out[0] if len(out) == 1 else out


def image_to_input(bgr, size=1024, device=None):
    """What the reference CLI feeds the network (run_ood_faceGAN_inversion.py:159-163): [0,255] BGR -> RGB NCHW in
    [-1,1], bilinearly resized (align_corners=False) to ``size`` if needed.  With a GPU ``device`` the image is uploaded
    at its own size and resized there by ``oodgan_resize_bilinear`` (the harness then runs no torch-ROCm kernel);
    without one the resize is the host-side ATen call the reference makes before its ``.cuda()``."""
    x = (img2tensor(bgr.astype(np.float64) / 255.0, bgr2rgb=True).unsqueeze(0) - 0.5) * 2
    if device is not None:
        x = x.to(device)
    if x.shape[-1] != size:
        if x.is_cuda:
            from . import samm
            x = samm.resize_bilinear(x.contiguous(), size)
        else:
            x = torch.nn.functional.interpolate(x, size=(size, size), mode='bilinear')
    return x


def extract_masks(aligns, size=1024, index=0):
    """aligns: dict level -> (B,3,H,W); channel 2 of every level, nearest-resized to ``size`` and concatenated along
    the width; returns the HxW*n uint8 strip of batch item ``index`` (the reference's CLI feeds one image: item 0; None on any
    failure, like the reference).  Device tensors go through ``oodgan_resize_nearest`` (one strip buffer, written in place); host tensors through ATen."""
    try:
        if all(v.is_cuda for v in aligns.values()):
            from . import samm
            return tensor2img(samm.extract_masks(aligns, size)[index], min_max=(0, 1))
        masks = []
        for k in sorted(aligns.keys()):
            m = aligns[k][:, 2:, ...]
            masks.append(torch.nn.functional.interpolate(m.float(), size=(size, size)))
        return tensor2img(torch.cat(masks, dim=3)[index], min_max=(0, 1))
    except Exception:
        return None


# ----------------------------------------------------------------------------------------------- metrics
def _prepare(img, img2, crop_border, input_order, test_y_channel):
    assert img.shape == img2.shape, f'Image shapes are different: {img.shape}, {img2.shape}.'
    if input_order not in ('HWC', 'CHW'):
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    out = []
    for a in (img, img2):
        a = np.asarray(a)
        if a.ndim == 2:
            a = a[..., None]
        elif input_order == 'CHW':
            a = a.transpose(1, 2, 0)
        a = a.astype(np.float64)
        if crop_border != 0:
            a = a[crop_border:-crop_border, crop_border:-crop_border, ...]
        if test_y_channel and a.shape[2] == 3:      # ITU-R BT.601 luma of a BGR image, [0,255] in and out
            # metric_util.py:32-46 + matlab_functions.py:236-244,353-359 with their roundings: float32 image / 255, the dot in
            # float64 (the coefficient list is a Python list), / 255 -> float32, x 255 in float32
            y = np.dot(a.astype(np.float32) / 255.0, [24.966, 128.553, 65.481]) + 16.0
            a = ((y / 255.0).astype(np.float32) * 255.0)[..., None]        # stays float32: PSNR's mean / log then run in float32
        out.append(a)
    return out


def calculate_psnr(img, img2, crop_border, input_order='HWC', test_y_channel=False, **kwargs):
    a, b = _prepare(img, img2, crop_border, input_order, test_y_channel)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return float(20.0 * np.log10(255.0 / np.sqrt(mse)))        # numpy scalar arithmetic in the arrays' dtype, as psnr_ssim.py:43-46


def _gauss_window(n=11, sigma=1.5):
    x = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _filter_valid(a, g):
    """Separable correlation with window g, valid region only (what filter2D(...)[5:-5, 5:-5] keeps)."""
    n = len(g)
    rows = sum(g[i] * a[i:a.shape[0] - n + 1 + i, :] for i in range(n))
    return sum(g[j] * rows[:, j:rows.shape[1] - n + 1 + j] for j in range(n))


def _ssim(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    g = _gauss_window()
    mu1, mu2 = _filter_valid(a, g), _filter_valid(b, g)
    s1 = _filter_valid(a * a, g) - mu1 * mu1
    s2 = _filter_valid(b * b, g) - mu2 * mu2
    s12 = _filter_valid(a * b, g) - mu1 * mu2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))).mean()


def calculate_ssim(img, img2, crop_border, input_order='HWC', test_y_channel=False, **kwargs):
    a, b = _prepare(img, img2, crop_border, input_order, test_y_channel)
    return float(np.mean([_ssim(a[..., i], b[..., i]) for i in range(a.shape[2])]))
