"""The metric's own horizon — "100 W+ steps" (BASELINE.json) — against the REAL reference: `Generator` autograd + `torch.optim.Adam`
run for 100 steps on the bench recipe (tests/golden/make_golden.py: gold_wplus_long; anchors model.py:483-585,
src/models/OOD_faceGAN_model.py:398-400, BasicSR losses.py:58-83) at 256² (images 0, 1) and at 1024² (image 10, image 0).

What a long trajectory can and cannot be compared on: Adam's update is lr * m / (sqrt(v) + eps), i.e. ~lr * sign(g) in the first steps,
so a coordinate whose gradient is ~0 takes either sign in two correct fp32 implementations and the latents drift apart coordinate-wise
while the LOSS CURVE — an average over 10^5 - 10^6 pixels of a function of all coordinates — stays together.  The fixtures therefore
hold the reference twice, in its own float32 and in float64 (same code through `G.double()`): their distance is the yardstick
("band") for how far two correct implementations are apart at a given step, and the tests require this build to be no further from the
float32 reference than a small multiple of that.  Steps 1-20: 1e-3 relative on every step's loss.

Things only a long run exercises, asserted here: the carried forward / backward range scales while w moves (no violation flag, no
re-run), Adam's bias correction at large t (the device step counter), the S-form-only saved activations over many overwrites, the
concurrent-stream split of a batch, and the two-instruction gradient path ('f16s-g2') staying inside the same band.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oodgan import synth  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _recipe(size, gidx, dev):
    cat = lambda parts: torch.cat(parts, 0).to(dev)
    target = cat([synth.make_images(size, 1, seed=1000 + g) for g in gidx])
    w0 = cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in gidx])
    per = [synth.make_noises(size, 1, seed=2000 + g) for g in gidx]
    noises = [cat([n[i] for n in per]) for i in range(len(per[0]))]
    return target, w0, noises


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _curve_report(tag, losses, ref, band=None):
    """max relative distance of the loss curves over steps [0,20), [20,60), [60,100) (whatever exists)."""
    out = {}
    n = min(losses.shape[0], ref.shape[0])
    rel = ((losses[:n] - ref[:n]).abs() / ref[:n].abs())
    for lo, hi in ((0, 20), (20, 60), (60, 100)):
        if lo < n:
            out[lo, hi] = float(rel[lo:min(hi, n)].max())
    msg = ', '.join(f'steps {lo + 1}-{min(hi, n)}: {v:.2e}' for (lo, hi), v in out.items())
    if band is not None:
        msg += ' | reference f32 vs its own f64: ' + ', '.join(f'{v:.2e}' for v in band.values())
    print(f'[{tag}] loss-curve rel distance to the reference: {msg}')
    return out


@pytest.mark.parametrize('prec', ['f16s', 'f16s-g2', 'f32'])
def test_100_steps_256_vs_reference_adam(dev, prec):
    from oodgan.engine import GeneratorEngine, WPlusInverter
    g32, g64 = _load('wplus_long_256.npz'), _load('wplus_long_256_f64.npz')
    size, gidx, steps = 256, [int(i) for i in g32['image_indices']], int(g32['steps'])
    assert steps == 100
    target, w0, noises = _recipe(size, gidx, dev)
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size, precision=prec)
    inv = WPlusInverter(eng)
    w, losses, traj = inv.invert(target, w0, noises, steps=steps, return_trajectory=True)
    assert inv.last_stats == {'steps_run': [steps], 'rollbacks': [0]}, f'range violation: {inv.last_stats}'
    assert not eng.bwd_scale_violated() and not eng.fwd_range_violated()
    L, L32, L64 = losses.double().cpu(), g32['losses'], g64['losses']
    band = _curve_report('reference f32 vs f64', L32, L64)
    d32 = _curve_report(f'256² {prec} vs reference f32', L, L32, band)
    d64 = _curve_report(f'256² {prec} vs reference f64', L, L64)
    # steps 1-20: 1e-3 on every step; later: within 4x the reference's own f32/f64 distance (floor 1e-3)
    assert d32[0, 20] < 1e-3 and d64[0, 20] < 1e-3
    for key in ((20, 60), (60, 100)):
        assert min(d32[key], d64[key]) < max(4 * band[key], 1e-3), (key, d32[key], d64[key], band[key])
    # the end of the metric's horizon: loss after 100 steps
    fin = abs(float(L[-1].mean()) - float(L32[-1].mean())) / float(L32[-1].mean())
    print(f'[256² {prec}] loss at step 100: {L[-1].tolist()} vs reference {L32[-1].tolist()} (mean rel {fin:.2e})')
    assert fin < 2e-3
    # latents: the bulk of the coordinates follows the reference (those with a clear gradient sign), the distance of the rest is
    # bounded by the step count times lr; reported next to the reference's own f32/f64 distance
    for t in (10, 30, 100):
        dw = (traj[t - 1].double().cpu() - g32[f'w_step{t}']).abs()
        dref = (g32[f'w_step{t}'] - g64[f'w_step{t}']).abs()
        print(f'[256² {prec}] w at step {t}: rms distance {float(dw.pow(2).mean().sqrt()):.2e} (reference f32 vs f64: {float(dref.pow(2).mean().sqrt()):.2e}), '
              f'within 2e-3: {float((dw < 2e-3).double().mean()):.4f} (reference: {float((dref < 2e-3).double().mean()):.4f})')
        assert float(dw.pow(2).mean().sqrt()) < max(3 * float(dref.pow(2).mean().sqrt()), 2e-3)
    assert torch.isfinite(w).all()


@pytest.mark.parametrize('prec', ['f16s', 'f16s-g2'])
@pytest.mark.parametrize('fixture', ['wplus_long_1024.npz', 'wplus_long_1024_img0.npz'])
def test_100_steps_1024_vs_reference_adam(dev, prec, fixture):
    """The benchmarked geometry, one image alone (the reference CLI's per-file mode, run_ood_faceGAN_inversion.py:158-182), the loop
    as bench.py runs it (device step counter, no trajectory)."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    if not os.path.exists(os.path.join(GOLDEN, fixture)):
        pytest.skip(f'{fixture} not generated')
    g32 = _load(fixture)
    if g32['losses'].shape[0] < 100:
        pytest.skip(f'{fixture} holds {g32["losses"].shape[0]} steps only (generation still running)')
    g64 = _load('wplus_long_1024_f64.npz') if fixture == 'wplus_long_1024.npz' and os.path.exists(os.path.join(GOLDEN, 'wplus_long_1024_f64.npz')) else None
    size, gidx = 1024, [int(i) for i in g32['image_indices']]
    steps = g32['losses'].shape[0]
    assert steps == 100
    target, w0, noises = _recipe(size, gidx, dev)
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size, precision=prec)
    inv = WPlusInverter(eng)
    w, losses = inv.invert(target, w0, noises, steps=steps)
    assert inv.last_stats == {'steps_run': [steps], 'rollbacks': [0]}, f'range violation: {inv.last_stats}'
    assert inv.last_plan['steps'] == [steps - 3] and inv.last_plan['launches'][0] >= 60, inv.last_plan     # steps 4..100 replayed from the launch plan
    assert not eng.bwd_scale_violated() and not eng.fwd_range_violated()
    L, L32 = losses.double().cpu(), g32['losses']
    band = _curve_report('reference f32 vs f64 (1024²)', L32, g64['losses']) if g64 is not None else None
    d32 = _curve_report(f'1024² {prec} vs reference f32 ({fixture})', L, L32, band)
    assert d32[0, 20] < 1e-3
    assert d32[20, 60] < 2e-3 and d32[60, 100] < 2e-3
    if band is not None:
        n64 = g64['losses'].shape[0]
        d64 = _curve_report(f'1024² {prec} vs reference f64', L[:n64], g64['losses'])
        assert d64[0, 20] < 1e-3
    fin = abs(float(L[-1]) - float(L32[-1])) / float(L32[-1])
    print(f'[1024² {prec}] loss at step 100: {float(L[-1]):.6f} vs reference {float(L32[-1]):.6f} (rel {fin:.2e})')
    assert fin < 1e-3
    for t in (10, 30, 100):
        dw = (w.double().cpu() - g32[f'w_step{t}']).abs() if t == 100 else None
        if dw is not None:
            print(f'[1024² {prec}] w at step 100: rms distance {float(dw.pow(2).mean().sqrt()):.2e}, within 2e-3: {float((dw < 2e-3).double().mean()):.4f}, '
                  f'moved from w0 by rms {float((g32["w_step100"] - w0.double().cpu()).pow(2).mean().sqrt()):.2e}')
            assert float(dw.pow(2).mean().sqrt()) < 0.25 * float((g32['w_step100'] - w0.double().cpu()).pow(2).mean().sqrt())
    # G(w_100) of THIS build against the reference's G(w_100): the images the two inversions end on
    img = eng.forward(w, noises)
    sub = img[:, :, ::16, ::16].double().cpu()
    e_img = float((sub - g32['final_image_sub'].double()).abs().max())
    e_rms = float((sub - g32['final_image_sub'].double()).pow(2).mean().sqrt())
    print(f'[1024² {prec}] G(w_100) vs the reference\'s: max |d pixel| {e_img:.2e}, rms {e_rms:.2e} (image std {float(g32["final_image_std"].mean()):.2f})')
    assert e_rms < 0.02 * float(g32['final_image_std'].mean())


def test_two_streams_and_one_stream_end_on_the_same_loss_b8(dev):
    """bench.py's default (two concurrent sub-batches of four) against the one-stream loop over the whole batch of 8 at 1024², 100
    steps: per-image final losses within 0.5 % — in fact bit-identical trajectories are not required (the 32² layers
    run another kernel at the sub-batch's size), the loss is."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, gidx = 1024, list(range(8))
    target, w0, noises = _recipe(size, gidx, dev)
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
    inv = WPlusInverter(eng)
    w1, l1 = inv.invert(target, w0, noises, steps=100, streams=1)
    w2, l2 = inv.invert(target, w0, noises, steps=100, streams=2)
    rel = ((l2[-1] - l1[-1]).abs() / l1[-1]).max().item()
    print(f'B=8 1024² 100 steps: final loss per image one stream {l1[-1].tolist()} | two streams: max rel distance {rel:.2e}; '
          f'mean {l1[-1].mean().item():.6f} / {l2[-1].mean().item():.6f}')
    assert rel < 5e-3
    assert ((l2[:20] - l1[:20]).abs() / l1[:20]).max().item() < 1e-3
    # image 0 of this batch against the reference's 100-step run of image 0 alone (per-image losses do not depend on the batch)
    if os.path.exists(os.path.join(GOLDEN, 'wplus_long_1024_img0.npz')) and _load('wplus_long_1024_img0.npz')['losses'].shape[0] == 100:
        ref = _load('wplus_long_1024_img0.npz')['losses'][:, 0]
        for tag, l in (('one stream', l1), ('two streams', l2)):
            r = ((l[:, 0].double().cpu() - ref).abs() / ref)
            print(f'image 0 in the batch of 8 ({tag}) vs the reference alone: steps 1-20 {float(r[:20].max()):.2e}, step 100 {float(r[-1]):.2e}')
            assert float(r[:20].max()) < 1e-3 and float(r[-1]) < 1e-3


@pytest.mark.parametrize('use_plan', [True, False])
@pytest.mark.parametrize('which', ['forward', 'backward'])
def test_range_violation_mid_run_repeats_one_window(dev, which, use_plan):
    """A carried scale sabotaged after step 15 of 40: the flags are read every 10 steps on a side stream, the run goes
    back to the last clean snapshot (step 10), repeats steps 11-20 with exact scales and continues in carry mode — 12 repeated steps
    (window + check lag) instead of all 40, same result as the undisturbed run.  With launch plans the plan of the flagged run is dropped
    and a new one recorded after the exact window."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B, steps = 32, 2, 40
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
    target = synth.make_images(size, B, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    w0 = synth.make_latents(size, B, seed=14).to(dev)
    inv = WPlusInverter(eng, use_plan=use_plan)
    w_ref, l_ref = inv.invert(target, w0, noises, steps=steps)
    assert inv.last_stats == {'steps_run': [steps], 'rollbacks': [0]}
    assert inv.last_plan['steps'] == [steps - 3 if use_plan else 0]
    done = {'n': 0}

    def sabotage(run):
        if run.steps_run == 15 and not done['n']:
            done['n'] = 1
            assert eng.carry_range and eng.fused_bwd
            if which == 'forward':
                eng.fwd_range.q[3].mul_(2.0 ** 12)
            else:
                next(iter(eng.bwd_state.values())).mul_(torch.tensor([2.0 ** 30, 2.0 ** -30], device=dev))
    inv.on_step = sabotage
    w, l = inv.invert(target, w0, noises, steps=steps)
    inv.on_step = None
    print(f'{which} scale sabotaged after step 15 of {steps} (plans {use_plan}): {inv.last_stats}, plan {inv.last_plan}')
    assert inv.last_stats['rollbacks'] == [1]
    assert inv.last_stats['steps_run'][0] == steps + inv.check_every + inv.check_lag
    assert eng.carry_range and eng.fused_bwd
    assert torch.isfinite(w).all()
    assert (l - l_ref).abs().max().item() <= 1e-5 * l_ref.abs().max().item()
    dw = (w - w_ref).abs()          # ten steps with exact instead of carried scales in the middle of 40 Adam steps: rounding-level differences, amplified
    assert (dw < 1e-3).float().mean().item() > 0.99 and dw.max().item() < 5e-3, (dw.max().item(), (dw < 1e-3).float().mean().item())
    # two concurrent sub-batches, undisturbed: no rollback on either stream
    w2, l2 = inv.invert(target, w0, noises, steps=steps, streams=2)
    assert inv.last_stats == {'steps_run': [steps, steps], 'rollbacks': [0, 0]}
    assert (l2 - l_ref).abs().max().item() <= 1e-4 * l_ref.abs().max().item()


@pytest.mark.parametrize('size,B,streams', [(64, 2, 1), (256, 4, 2), (1024, 8, 2), (1024, 1, 1)])
def test_launch_plan_is_bit_identical_to_the_python_driven_loop(dev, size, B, streams):
    """steps 4..N re-issued from the recorded launch plan (oodgan_plan_run: ~170 launches per call) against the same
    loop driven launch by launch from Python — same kernels, same arguments, same order: bit-identical latents and losses."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    steps = 12
    target, w0, noises = _recipe(size, list(range(B)), dev)
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
    inv_p, inv_e = WPlusInverter(eng, use_plan=True), WPlusInverter(eng, use_plan=False)
    w_e, l_e = inv_e.invert(target, w0, noises, steps=steps, streams=streams)
    w_p, l_p = inv_p.invert(target, w0, noises, steps=steps, streams=streams)
    w_p2, l_p2 = inv_p.invert(target, w0, noises, steps=steps, streams=streams)       # a second inversion records its own plans
    print(f'{size}² B={B} streams={streams}: plan {inv_p.last_plan}, eager {inv_e.last_plan}')
    assert inv_p.last_plan['steps'] == [steps - 3] * streams and all(n > 60 for n in inv_p.last_plan['launches'])
    assert inv_e.last_plan['steps'] == [0] * streams
    assert torch.equal(l_p, l_e) and torch.equal(w_p, w_e)
    assert torch.equal(l_p2, l_e) and torch.equal(w_p2, w_e)


def test_empty_batch_and_zero_steps(dev):
    """Edge cases of the loop's interface: an empty batch (a rank whose shard is empty) and steps = 0 return the start latents and an empty loss
    table without launching anything."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size = 32
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
    noises = [n.to(dev) for n in synth.make_noises(size, 2, seed=7)]
    inv = WPlusInverter(eng)
    w, l = inv.invert(torch.empty(0, 3, size, size, device=dev), torch.empty(0, eng.n_latent, 512, device=dev), [n[:0] for n in noises], steps=5, streams=2)
    assert w.shape == (0, eng.n_latent, 512) and l.shape == (0, 0)
    w0 = synth.make_latents(size, 2, seed=14).to(dev)
    w, l, traj = inv.invert(synth.make_images(size, 2, seed=9).to(dev), w0, noises, steps=0, return_trajectory=True)
    assert torch.equal(w, w0) and l.shape == (0, 2) and traj == []


def test_launch_plan_in_a_fresh_process_state(dev):
    """The hazard the second Python-driven step exists for (engine._WRun): a persistent scratch buffer first created INSIDE the recorded step is carved
    out of blocks its temporaries used earlier and gets overwritten on every replay.  A fresh engine + the hi-only record buffers (created only by
    the carried-scale path) + plan + range guard, from the very first inversion: no rollback, same result as the Python-driven loop."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    from oodgan import ops
    size, B, steps = 256, 4, 16
    target, w0, noises = _recipe(size, list(range(B)), dev)
    ops._SFORM_POOL.clear()                 # as in a fresh process: every pooled scratch buffer is created by this inversion
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
    inv = WPlusInverter(eng, use_plan=True)
    w_p, l_p = inv.invert(target, w0, noises, steps=steps)
    assert inv.last_stats == {'steps_run': [steps], 'rollbacks': [0]} and inv.last_plan['steps'] == [steps - 3], (inv.last_stats, inv.last_plan)
    assert torch.isfinite(l_p).all()
    w_e, l_e = WPlusInverter(eng, use_plan=False).invert(target, w0, noises, steps=steps)
    assert torch.equal(l_p, l_e) and torch.equal(w_p, w_e)


@pytest.mark.parametrize('size,B,streams', [(256, 1, 1), (256, 3, 2), (256, 5, 3), (512, 2, 1), (512, 5, 2), (1024, 3, 2)])
def test_ragged_batches_plan_and_hi_records_vs_three_instruction_loop(dev, size, B, streams):
    """Batch sizes and stream counts off the benchmarked 8 / 2 (ragged sub-batches take other kernels per sub-batch: the 8-wave kernels and with them
    the two-instruction instances and the hi-only records start at 128 — 64 with concurrent streams — work items): the default loop (f16s-g2, launch plans,
    range guard) is bit-identical to its Python-driven self and within the gradient-rounding band of the three-instruction loop ('f16s')."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    steps = 10
    target, w0, noises = _recipe(size, list(range(B)), dev)
    state = {k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}
    eng = GeneratorEngine(state, size)
    assert eng.grad_hi_only
    inv = WPlusInverter(eng)
    w_p, l_p = inv.invert(target, w0, noises, steps=steps, streams=streams)
    ns = min(streams, B)
    assert inv.last_stats == {'steps_run': [steps] * ns, 'rollbacks': [0] * ns} and inv.last_plan['steps'] == [steps - 3] * ns, (inv.last_stats, inv.last_plan)
    w_e, l_e = WPlusInverter(eng, use_plan=False).invert(target, w0, noises, steps=steps, streams=streams)
    assert torch.equal(l_p, l_e) and torch.equal(w_p, w_e)
    eng3 = GeneratorEngine(state, size, precision='f16s')
    w_3, l_3 = WPlusInverter(eng3).invert(target, w0, noises, steps=steps, streams=streams)
    rel = ((l_p - l_3).abs() / l_3).max().item()
    dw = (w_p - w_3).abs()
    print(f'{size}² B={B} streams={streams}: f16s-g2 vs f16s over {steps} steps: loss rel {rel:.2e}, |dw| max {dw.max().item():.2e}, within 2e-3: {(dw < 2e-3).float().mean().item():.5f}')
    assert rel < 1e-3 and (dw < 2e-3).float().mean().item() > 0.995
