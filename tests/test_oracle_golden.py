"""Pin the CPU oracle (oracle/ref_cpu.py) against vectors produced by the real reference
(tests/golden/make_golden.py).  Runs on CPU, no GPU, no /root/reference."""
import torch
import pytest

from oracle import ref_cpu as R
from oodgan import synth

TOL = 1e-5


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, f'max abs err {err:.3e} (ref max {ref:.3e})'


def test_upfirdn2d_modes(golden):
    g = golden('ops.npz')
    k4 = R.make_kernel([1, 3, 3, 1])
    x = g['ufd_x']
    for tag, kern, up, down, pad in [
        ('blur11', k4 * 4, 1, 1, (1, 1)), ('up2', k4 * 4, 2, 1, (2, 1)), ('blur21', k4, 1, 1, (2, 1)),
        ('down2', k4 * 4, 1, 2, (1, 2)), ('blur22', k4 * 4, 1, 1, (2, 2)), ('crop', k4, 1, 1, (-1, 3)),
    ]:
        close(R.upfirdn2d(x, kern, up, down, pad), g[f'ufd_{tag}'])


def test_fused_leaky_relu_and_linear(golden):
    g = golden('ops.npz')
    close(R.fused_leaky_relu(g['ufd_x'], g['flr_b']), g['flr_y'])
    close(R.fused_leaky_relu(g['ufd_x'], g['flr_b'], 0.1, 1.5), g['flr_y2'])
    close(R.equal_linear(g['lin_x'], g['lin_w'], g['lin_b'], 0.01, False), g['lin_y'])
    close(R.equal_linear(g['lin_x'], g['lin_w'], g['lin_b'], 0.01, True), g['lin_y_act'])


def test_modulated_conv_variants(golden):
    g = golden('ops.npz')
    for tag, demod, ups in [('plain', True, False), ('up', True, True), ('rgb', False, False)]:
        y = R.modulated_conv2d(g['mc_x'], g['mc_wlat'], g[f'mc_{tag}_w'], g[f'mc_{tag}_mw'], g[f'mc_{tag}_mb'],
                               demod, ups)
        close(y, g[f'mc_{tag}_y'])


def test_styled_conv_and_to_rgb(golden):
    g = golden('ops.npz')
    B, Ci, Co, H, S = 2, 16, 8, 12, 64
    for tag, ups in [('sc', False), ('scup', True)]:
        P = {}
        synth._styled_conv(P, 'q', Ci, Co, S, 13, ups, 0.1)
        close(R.styled_conv(P, 'q', g['mc_x'], g['mc_wlat'], g[f'{tag}_noise'], ups), g[f'{tag}_y'])
    P = {}
    synth._to_rgb(P, 'q', Ci, S, 13, True)
    close(R.to_rgb(P, 'q', g['mc_x'], g['mc_wlat'], g['rgb_skip']), g['rgb_y'])
    close(R.to_rgb(P, 'q', g['mc_x'], g['mc_wlat'], None), g['rgb_y_noskip'])


def test_generator_s32(golden):
    g = golden('generator_s32.npz')
    P = synth.generator_state(32, seed=5)
    lat = synth.make_latents(32, 2, seed=6)
    noises = synth.make_noises(32, 2, seed=7)
    with torch.no_grad():
        img, feats = R.generator_forward(P, lat, noises, 32, return_features=True)
        close(img, g['image'])
        close(feats[-1][:, ::16], g['last_feature_sub'])
        # mapping network + stored noise buffers
        w = R.mapping_network(P, g['z'])
        close(w.unsqueeze(1).repeat(1, 8, 1), g['latent_from_z'], 2e-5)
        stored = [P[f'noises.noise_{i}'] for i in range(7)]
        close(R.generator_forward(P, w.unsqueeze(1).repeat(1, 8, 1), stored, 32), g['image_from_z'], 5e-5)
        wt = g['mean_lat'] + 0.7 * (w - g['mean_lat'])
        close(R.generator_forward(P, wt.unsqueeze(1).repeat(1, 8, 1), stored, 32), g['image_trunc'], 5e-5)


def test_wplus_trajectory_s32(golden):
    g = golden('wplus_s32.npz')
    P = synth.generator_state(32, seed=5)
    target = synth.make_images(32, 2, seed=9)
    noises = synth.make_noises(32, 2, seed=7)
    w0 = synth.make_latents(32, 2, seed=6)
    w, losses, traj = R.wplus_invert(P, target, w0, noises, 32, steps=5, return_trajectory=True)
    close(losses, g['losses'], 1e-4)
    close(torch.stack(traj), g['traj'], 1e-4)


def test_wplus_trajectory_256(golden):
    """the oracle's W+ loop at 256² against the reference Generator + torch.optim.Adam trajectory."""
    g = golden('wplus_256.npz')
    size, B = 256, 2
    P = synth.generator_state(size, seed=0)
    w, losses, traj = R.wplus_invert(P, synth.make_images(size, B, seed=71), synth.make_latents(size, B, seed=73, std=0.3),
                                     synth.make_noises(size, B, seed=72), size, steps=5, return_trajectory=True)
    close(losses, g['losses'], 1e-4)
    dw = (torch.stack(traj) - g['traj']).abs()
    assert (dw < 2e-3).float().mean().item() > 0.999, dw.max().item()


def test_wplus_step_1024_oracle_vs_reference(golden):
    """the oracle at the benchmarked geometry (1024², bench recipe image 10): loss and dL/dW+ of one step in the oracle's
    fp32 vs the reference Generator evaluated in float64 (the reference's own fp32 is 2.5e-5 from it)."""
    g = golden('wplus_1024.npz')
    size, gi = 1024, int(g['image_index'])
    P = synth.generator_state(size, seed=0)
    w = synth.make_latents(size, 1, seed=3000 + gi, std=0.3).requires_grad_(True)
    img = R.generator_forward(P, w, synth.make_noises(size, 1, seed=2000 + gi), size)
    loss = R.wplus_loss(img, synth.make_images(size, 1, seed=1000 + gi))
    loss.backward()
    assert abs(loss.item() - g['loss_f64'].item()) < 1e-5 * g['loss_f64'].item()
    assert (img.detach()[:, :, ::16, ::16] - g['image_sub']).abs().max().item() < 1e-4
    rel = (w.grad.double() - g['grad_f64']).abs().max().item() / g['grad_f64'].abs().max().item()
    assert rel < 2e-4, rel


def test_samm(golden):
    g = golden('samm.npz')
    P = synth.samm_state(8, 'm', seed=21)
    with torch.no_grad():
        close(R.align_net(P, 'm.alignment.body', g['tgt'], g['src'], 0.08), g['alignnet'])
        y, f = R.spm_warp(P, 'm.alignment', g['src'], g['tgt'], None, 0.08, 2)
        close(y, g['warp_out'])
        close(f, g['warp_field'])
        y, f = R.spm_warp(P, 'm.alignment', g['src'], g['tgt'], g['prev'], 0.08, 2)
        close(y, g['warp_out_prev'])
        close(f, g['warp_field_prev'])
        y, f = R.spm_warp(P, 'm.alignment', g['src'], g['tgt'], g['prev'], 0.08, 1)
        close(y, g['warp1_out_prev'])
        close(f, g['warp1_field_prev'])
        close(R.new_prm(g['prev'][:, 2:], g['warp_field'][:, 2:]), g['prm_up'])
        close(R.new_prm(g['warp_field'][:, 2:], g['warp_field_prev'][:, 2:]), g['prm_same'])


def test_samm_without_the_difference_input(golden):
    """AlignNet / SPM_Warp with diff_fAndg=False (reference SAMM/helpers.py:98-101) against vectors of the real modules."""
    g, g0 = golden('samm_nodiff.npz'), golden('samm.npz')
    P = synth.samm_state(8, 'm', seed=21)
    with torch.no_grad():
        close(R.align_net(P, 'm.alignment.body', g0['tgt'], g0['src'], 0.08, diff_fAndg=False), g['alignnet'])
        y, f = R.spm_warp(P, 'm.alignment', g0['src'], g0['tgt'], g0['prev'], 0.08, 2, diff_fAndg=False)
        close(y, g['warp_out_prev'])
        close(f, g['warp_field_prev'])
        assert (g['alignnet'] - g0['alignnet']).abs().max() > 1e-3          # the option changes the result


def test_ood_forward_1024(golden):
    """Full 1024² OOD forward after the encoder (≈15-20 s on 8 cores)."""
    g = golden('ood_1024.npz')
    P = synth.ood_state(1024, seed=31)
    enc_lats = synth.make_latents(1024, 1, seed=32, std=0.3)
    enc_feats = synth.make_encoder_feats(1, seed=33)
    x = synth.make_images(1024, 1, seed=34)
    noises = synth.make_noises(1024, 1, seed=35)
    with torch.no_grad():
        out, lats, aligns = R.ood_forward(P, x, enc_lats, enc_feats, noises)
    close(lats, g['lats'])
    tol = 2e-4  # different thread partitioning of the same ATen graph (2.7e-6 rel / layer)
    close(out[:, :, ::16, ::16], g['out_sub'], tol)
    close(out[:, :, 480:544, 480:544], g['out_crop'], tol)
    close(out.mean(dim=(2, 3)), g['out_mean'], tol)
    for k in (1, 2, 3, 4):
        a = aligns[k]
        step = max(1, a.shape[-1] // 32)
        close(a[:, :, ::step, ::step], g[f'align{k}_sub'], tol)
    close(aligns[1024][:, :, ::16, ::16], g['align1024_sub'], tol)
    strip = R.extract_masks(aligns)
    close(strip[:, :, ::16, ::16], g['mask_strip_sub'], tol)
    close(strip[:, :, 500:502, :], g['mask_strip_rows'], tol)


def test_cond_types_vs_reference(golden):
    """feature_modulation conditioning (cond_type 'SFT' / 'ADD' / 'FUSE', model.py:558-566,588-610) of the oracle."""
    g = golden('cond_types_s32.npz')
    P = synth.generator_state(32, seed=5)
    B = 2
    lat, noises = synth.make_latents(32, B, seed=6), synth.make_noises(32, B, seed=7)
    conds = [[synth.normal(f'cond.{k}.0', (B, 512, r, r), 9, 0.5), synth.normal(f'cond.{k}.1', (B, 512, r, r), 10, 0.5)] for k, r in ((0, 8), (1, 16))]
    with torch.no_grad():
        for ct in ('SFT', 'ADD', 'FUSE'):
            img, feats = R.generator_forward(P, lat, noises, 32, cond_layers=[1, 3], return_features=True,
                                             post_hook=lambda k, out, ct=ct: R.feature_modulation(out, conds[k], ct))
            close(img, g[f'image_{ct}'], 1e-4)
            close(feats[-1][:, ::16], g[f'feat_{ct}_sub'], 1e-4)


def test_generator_1024_b4_oracle_vs_reference(golden):
    """BASELINE configs[1] (C2): the oracle's mapping network + 1024² generator forward on image 2 of the reference's batch of 4
    (generator_1024_b4.npz, produced by the reference Generator in fp32; one image keeps the CPU suite short — images of a
    batch are independent in every op of the path)."""
    from make_golden_params import GEN_B4
    g = golden('generator_1024_b4.npz')
    size, B, k = GEN_B4['size'], GEN_B4['batch'], 2
    P = synth.generator_state(size, seed=0)
    z = synth.normal('gen_b4.z', (B, 512), GEN_B4['z_seed'])
    noises = [n[k:k + 1] for n in synth.make_noises(size, B, seed=GEN_B4['noise_seed'])]
    with torch.no_grad():
        w = R.mapping_network(P, z)
        close(w, g['latent'], 2e-5)
        img = R.generator_forward(P, w[k:k + 1].unsqueeze(1).repeat(1, 18, 1), noises, size)
    close(img[:, :, ::16, ::16], g['image_sub'][k:k + 1], 5e-5)
    close(img[:, :, 448:512, 512:576], g['image_crop'][k:k + 1], 5e-5)
    assert (img.double().mean(dim=(2, 3)) - g['image_mean'][k:k + 1]).abs().max().item() < 1e-5


@pytest.mark.parametrize('size', [64, 256])
def test_wplus_asymmetric_blur_kernel_oracle_vs_reference(golden, size):
    """``Generator(blur_kernel=[1,4,2,1])`` (model.py:376-384): the oracle takes Blur / Upsample kernels from the state dict, as the reference's
    registered buffers do; the asymmetric taps pin every flip of the chain.  64²: ToRGB's Upsample as constructed ([1,3,3,1], model.py:455);
    256²: a second asymmetric kernel there (checkpoint buffers).  Image, loss curve and dL/dW+ vs the reference in float64."""
    g = golden(f'wplus_blur_{size}.npz')
    gidx = [int(i) for i in g['image_indices']]
    P = synth.generator_state(size, seed=0, blur_kernel=tuple(int(t) for t in g['taps']), upsample_kernel=tuple(int(t) for t in g['up_taps']))
    cat = lambda parts: torch.cat(parts, 0)
    target = cat([synth.make_images(size, 1, seed=1000 + i) for i in gidx])
    per = [synth.make_noises(size, 1, seed=2000 + i) for i in gidx]
    noises = [cat([n[k] for n in per]) for k in range(len(per[0]))]
    w0 = cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in gidx])
    w = w0.clone().requires_grad_(True)
    img = R.generator_forward(P, w, noises, size)
    R.wplus_loss(img, target).backward()
    st = max(size // 64, 1)
    assert (img.detach()[:, :, ::st, ::st] - g['image_sub']).abs().max().item() < 1e-4
    rel = (w.grad.double() - g['grad_f64']).abs().max().item() / g['grad_f64'].abs().max().item()
    assert rel < 1e-4, rel
    steps = g['losses'].shape[0]
    _, losses, traj = R.wplus_invert(P, target, w0, noises, size, steps=steps, return_trajectory=True)
    close(losses.double(), g['losses'], 1e-4)
    assert ((torch.stack(traj).double() - g['traj']).abs() < 2e-3).float().mean().item() > 0.999
    # the default-kernel state must NOT reproduce it (the fixture really depends on the taps)
    with torch.no_grad():
        img0 = R.generator_forward(synth.generator_state(size, seed=0), w0, noises, size)
    assert (img0[:, :, ::st, ::st] - g['image_sub']).abs().max().item() > 1e-2
