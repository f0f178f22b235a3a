"""CPU checks around the LPIPS term (no GPU): properties of the oracle restatement (oracle/lpips_cpu.py — PARITY UNPINNED, the `lpips`
package is absent: SURVEY.md §8c) and the host-side weight transforms of oodgan/lpips.py, which are plain torch and can be pinned here:
AlexNet's 11x11 stride-4 conv as a 3x3 stride-1 VALID conv over the 4x4 space-to-depth image, and the packed layouts of
oodgan_conv2d_s1 (forward and flipped / transposed for the input gradient)."""
import torch
import torch.nn.functional as F

from oracle import lpips_cpu as LO
from oodgan import synth


def test_oracle_properties():
    P = {k: v.double() for k, v in synth.lpips_state(0).items()}
    a = synth.make_images(64, 2, seed=1).double()
    b = synth.make_images(64, 2, seed=2).double()
    _, dab = LO.lpips_loss(P, a, b, min_max=(-1, 1), reduction='none')
    _, dba = LO.lpips_loss(P, b, a, min_max=(-1, 1), reduction='none')
    _, daa = LO.lpips_loss(P, a, a, min_max=(-1, 1), reduction='none')
    assert (dab > 0).all() and torch.allclose(dab, dba, rtol=1e-12) and float(daa.abs().max()) == 0.0
    # range handling of the reference wrapper (lpips_loss.py:27-31): min_max (0,1) on x01 == min_max (-1,1) on 2*x01-1
    _, d01 = LO.lpips_loss(P, a * 0.5 + 0.5, b * 0.5 + 0.5, min_max=(0, 1), reduction='none')
    assert torch.allclose(d01, dab, rtol=1e-10)
    l, _ = LO.lpips_loss(P, a, b, loss_weight=0.5, min_max=(-1, 1))
    assert abs(float(l) - 0.5 * float(dab.mean())) < 1e-14
    ar = a.clone().requires_grad_(True)
    LO.lpips_loss(P, ar, b, min_max=(-1, 1))[0].backward()
    assert torch.isfinite(ar.grad).all() and float(ar.grad.abs().max()) > 0
    taps = LO.alexnet_taps(P, a)
    assert [t.shape[1] for t in taps] == list(LO.CHANNELS) and [t.shape[2] for t in taps] == [15, 7, 3, 3, 3]


def test_conv1_as_3x3_over_space_to_depth_and_packed_layouts():
    from oodgan.lpips import _conv1_as_3x3, _pack
    w = synth.normal('t.w1', (64, 3, 11, 11), 3, 0.05).double()
    x = synth.normal('t.x', (2, 3, 64, 96), 4).double()
    ref = F.conv2d(x, w, stride=4, padding=2)
    xp = F.pad(x, (2, 2, 2, 2))
    B, _, Hp, Wp = xp.shape
    s2d = xp.view(B, 3, Hp // 4, 4, Wp // 4, 4).permute(0, 1, 3, 5, 2, 4).reshape(B, 48, Hp // 4, Wp // 4)       # channel c*16 + dy*4 + dx
    w3 = _conv1_as_3x3(w)
    out = F.conv2d(s2d, w3)
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 1e-12
    # packed layouts: [K][tap][Mp]
    wf, wb = _pack(w3.float(), False), _pack(w3.float(), True)
    assert wf.shape == (48, 9, 64) and wb.shape == (64, 9, 64)
    assert torch.equal(wf[5, 4, :64], w3.float()[:, 5, 1, 1]) and torch.equal(wb[7, 0, :48], w3.float()[7, :, 2, 2]) and float(wb[:, :, 48:].abs().max()) == 0.0
    # input gradient = conv with the flipped / transposed kernel and pad' = ks-1-pad
    g = synth.normal('t.g', tuple(out.shape), 5).double()
    s2dr = s2d.clone().requires_grad_(True)
    (dx,) = torch.autograd.grad(F.conv2d(s2dr, w3), s2dr, g)
    wt = torch.flip(w3, [2, 3]).transpose(0, 1)
    assert float((F.conv2d(g, wt, padding=2) - dx).abs().max()) < 1e-12
