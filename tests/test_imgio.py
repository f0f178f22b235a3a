"""Image I/O + metrics harness: uint8 rounding, channel order, PSNR closed form, SSIM against a
direct double-loop evaluation of the published formula, nearest-neighbour mask strip against the oracle."""
import math

import numpy as np
import pytest
import torch

from oodgan import imgio


def test_img_tensor_roundtrip_and_rounding(tmp_path):
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (12, 9, 3), dtype=np.uint8)
    t = imgio.img2tensor(bgr.astype(np.float64) / 255.0)
    assert t.shape == (3, 12, 9) and t.dtype == torch.float32
    assert torch.equal(t[0], torch.from_numpy(bgr[:, :, 2].astype(np.float32) / 255.0))       # channel 0 is R
    assert np.array_equal(imgio.tensor2img(t), bgr)                                           # exact round trip
    # x255 then ROUND (not truncate): 0.5/255 below a half step stays, above goes up
    v = torch.tensor([[[0.4999 / 255, 0.5001 / 255, 254.5001 / 255, 2.0, -1.0]]])
    assert imgio.tensor2img(v).tolist() == [[0, 1, 255, 255, 0]]
    x = torch.linspace(-1.2, 1.2, 7).reshape(1, 1, 1, 7)
    ref = ((x.clamp(-1, 1) + 1) / 2 * 255).round().numpy().astype(np.uint8).reshape(1, 7)
    assert np.array_equal(imgio.tensor2img(x, min_max=(-1, 1)), ref)
    p = tmp_path / 'a' / 'x.png'
    imgio.imwrite(str(p), bgr)
    assert np.array_equal(imgio.imread(str(p)), bgr)
    inp = imgio.image_to_input(bgr.astype(np.float64), size=12) if bgr.shape[0] == bgr.shape[1] else imgio.image_to_input(bgr.astype(np.float64), size=16)
    assert inp.shape[0] == 1 and inp.shape[1] == 3 and inp.abs().max() <= 1.0


def test_psnr_closed_form_and_crop():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (20, 24, 3)).astype(np.float64)
    b = np.clip(a + rng.normal(0, 5, a.shape), 0, 255)
    c = 2
    mse = np.mean((a[c:-c, c:-c] - b[c:-c, c:-c]) ** 2)
    assert imgio.calculate_psnr(a, b, crop_border=c) == pytest.approx(20 * math.log10(255 / math.sqrt(mse)), rel=1e-12)
    assert imgio.calculate_psnr(a, a, crop_border=0) == float('inf')
    assert imgio.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), crop_border=c, input_order='CHW') == \
        pytest.approx(imgio.calculate_psnr(a, b, crop_border=c))
    # Y channel (BT.601 on BGR): a pure blue change moves Y by 24.966/255 per level
    d = a.copy()
    d[..., 0] = np.clip(d[..., 0] + 10, 0, 255)
    y = imgio.calculate_psnr(a, d, crop_border=0, test_y_channel=True)
    dy = (d[..., 0] - a[..., 0]) * 24.966 / 255.0
    assert y == pytest.approx(20 * math.log10(255 / math.sqrt(np.mean(dy ** 2))), rel=1e-4)
    with pytest.raises(ValueError):
        imgio.calculate_psnr(a, b, 0, input_order='WHC')


def _ssim_direct(a, b):
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    w = np.outer(g / g.sum(), g / g.sum())
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for i in range(a.shape[0] - 10):
        for j in range(a.shape[1] - 10):
            pa, pb = a[i:i + 11, j:j + 11], b[i:i + 11, j:j + 11]
            m1, m2 = (w * pa).sum(), (w * pb).sum()
            s1, s2, s12 = (w * pa * pa).sum() - m1 * m1, (w * pb * pb).sum() - m2 * m2, (w * pa * pb).sum() - m1 * m2
            vals.append(((2 * m1 * m2 + c1) * (2 * s12 + c2)) / ((m1 * m1 + m2 * m2 + c1) * (s1 + s2 + c2)))
    return float(np.mean(vals))


def test_ssim_matches_direct_evaluation():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (26, 30, 3)).astype(np.float64)
    b = np.clip(a + rng.normal(0, 12, a.shape), 0, 255)
    ref = np.mean([_ssim_direct(a[2:-2, 2:-2, i], b[2:-2, 2:-2, i]) for i in range(3)])
    assert imgio.calculate_ssim(a, b, crop_border=2) == pytest.approx(ref, rel=1e-10)
    assert imgio.calculate_ssim(a, a, crop_border=0) == pytest.approx(1.0, abs=1e-12)
    assert 0 < imgio.calculate_ssim(a, b, crop_border=0) < 1


def test_mask_strip_matches_oracle():
    g = torch.Generator().manual_seed(3)
    aligns = {k: torch.rand(1, 3, 8 * 2 ** k, 8 * 2 ** k, generator=g) for k in (1, 2, 3)}
    aligns[64] = torch.rand(1, 1, 64, 64, generator=g).repeat(1, 3, 1, 1)
    strip = imgio.extract_masks(aligns, size=64)
    cols = [torch.nn.functional.interpolate(aligns[k][:, 2:], size=(64, 64)) for k in sorted(aligns)]
    exp = (torch.cat(cols, dim=3)[0, 0].clamp(0, 1) * 255).round().numpy().astype(np.uint8)
    assert strip.shape == (64, 64 * 4) and np.array_equal(strip, exp)
    assert imgio.extract_masks({1: torch.rand(2)}) is None            # any failure -> None, like the reference


def test_tensor2img_img2tensor_psnr_vs_reference_vectors(golden):
    """Vectors produced by the REAL BasicSR functions (tests/golden/make_golden.py gold_imgio): the uint8 rounding rule of
    ``tensor2img`` (x255 then numpy round = half to even, after the clamp), its float and batch-of-one paths, the
    single-channel mask path of the CLI, ``img2tensor`` and ``calculate_psnr`` (crop, CHW order, BT.601 Y channel of a
    BGR image, uint8 and float inputs), and ``calculate_ssim`` from the real psnr_ssim.py:49-128 with numpy stand-ins for its two
    cv2 primitives (getGaussianKernel / filter2D, see make_golden.py).  The channel swap itself (cv2.cvtColor) cannot be run here
    — cv2 is absent — and stays unpinned."""
    import numpy as np
    z = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'imgio.npz'))
    t = torch.from_numpy(z['t'])
    assert np.array_equal(imgio.tensor2img(t.clone(), rgb2bgr=False, min_max=(-1, 1)), z['t2i_rgb_u8'])
    assert np.array_equal(imgio.tensor2img(t.clone(), rgb2bgr=False, out_type=np.float32, min_max=(-1, 1)), z['t2i_rgb_f32'])
    assert np.array_equal(imgio.tensor2img(t.clone().unsqueeze(0), rgb2bgr=False, min_max=(-1, 1)), z['t2i_batch1_u8'])
    assert np.array_equal(imgio.tensor2img(t.clone(), rgb2bgr=True, min_max=(-1, 1)), z['t2i_rgb_u8'][:, :, ::-1])
    assert np.array_equal(imgio.tensor2img(torch.from_numpy(z['mask']), min_max=(0, 1)), z['t2i_mask_u8'])
    got = imgio.img2tensor(z['img_f64'].copy(), bgr2rgb=False, float32=True)
    assert got.dtype == torch.float32 and torch.equal(got, torch.from_numpy(z['i2t']))
    a, b, v = z['psnr_a'], z['psnr_b'], z['psnr_vals']
    mine = [imgio.calculate_psnr(a, b, crop_border=0), imgio.calculate_psnr(a, b, crop_border=4),
            imgio.calculate_psnr(a, b, crop_border=4, test_y_channel=True),
            imgio.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), crop_border=2, input_order='CHW'),
            imgio.calculate_psnr(a.astype(np.float64), b.astype(np.float64), crop_border=0, test_y_channel=True)]
    assert np.allclose(mine, v, rtol=1e-12, atol=0), (mine, v.tolist())
    sv = z['ssim_vals']
    ssim = [imgio.calculate_ssim(a, b, crop_border=0), imgio.calculate_ssim(a, b, crop_border=4),
            imgio.calculate_ssim(a, b, crop_border=4, test_y_channel=True),
            imgio.calculate_ssim(a.transpose(2, 0, 1), b.transpose(2, 0, 1), crop_border=2, input_order='CHW'),
            imgio.calculate_ssim(a, a, crop_border=0)]
    assert np.allclose(ssim, sv, rtol=1e-10, atol=0), (ssim, sv.tolist())
    assert 0.5 < sv[0] < 1.0 and sv[4] == pytest.approx(1.0, abs=1e-12)
