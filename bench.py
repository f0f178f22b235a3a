#!/usr/bin/env python3
"""Headline benchmark: 1024² face inversions/sec (100 W+ Adam steps + final OOD/SAIM forward).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N=1 directly; N>1 self-launches N rank processes
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: B=8 images per GPU (BASELINE.json configs[2]:
"Full OOD inversion loop (100 W+ Adam steps + SAIM mask) 1024² batch=8, 1xMI355X"), i.e. 100 x
{generator forward, per-image MSE, backward to W+, Adam} followed by one full OOD forward (SAMM
alignment + invertibility masks + blend) with the refined latents.  Inputs (weights, target
images, encoder latents/features, noise maps) are synthetic (oodgan.synth) and resident in HBM
before the timed region.  Multi-GPU = batch sharding (weak scaling): every rank inverts its own 8
images, the only collective is one all_gather of the finished latents (RCCL over xGMI).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events on
the launch stream) and `cpu_baseline` (the CPU oracle on the host cores, bounded sample)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'ood-gan-inversion_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ != '__main__':
    import torch  # noqa: E402  (imported by tools/ as a module: no launcher decision to make; main() imports it after that decision)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = vector peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA (spec, MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    ap.add_argument('--wsteps', type=int, default=100, help='W+ Adam steps per inversion (metric is quoted at 100)')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline-events', action='store_true')
    ap.add_argument('--no-modconv', action='store_true', help='skip the fp16 modulated-conv roofline leg (BASELINE configs[4])')
    ap.add_argument('--streams', type=int, default=2, help='independent sub-batches of the per-GPU batch advanced on separate HIP streams (DESIGN.md §10)')
    ap.add_argument('--roofline-steps', type=int, default=10, help='W+ steps of the exclusive single-stream pass that times the dominant kernel')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the extra leg that times the inversion including the e4e encoder')
    ap.add_argument('--no-forward-only', action='store_true', help="skip the leg that times the reference's own path: encoder + OOD forward, no W+ steps")
    ap.add_argument('--no-generator-fwd', action='store_true', help='skip the BASELINE configs[1] leg: plain generator forward at batch 4')
    ap.add_argument('--no-single-stream', action='store_true', help='skip the extra leg that times the same job on ONE HIP stream')
    ap.add_argument('--graph', type=int, default=0, help='1: replay each W+ step from a captured hipGraph')
    ap.add_argument('--precision', default='f16s-g2', choices=['f16s-g2', 'f16s', 'f32'],
                    help="conv arithmetic (DESIGN.md §2): 'f16s-g2' = split-f16, 3 matrix instructions per product in the forward and 2 in the input-gradient "
                         "convs (default since round 6: 100-step loss curve and dL/dW+ inside the same band as 'f16s'); 'f16s' = 3 everywhere; 'f32' = exact fp32 MFMA")
    ap.add_argument('--no-plan', action='store_true', help='drive every W+ step launch by launch from Python instead of replaying the recorded launch plan (oodgan_plan_*)')
    ap.add_argument('--no-b1', action='store_true', help="skip the B = 1 inversion leg (the reference CLI's per-file mode)")
    ap.add_argument('--no-lpips', action='store_true', help='skip the leg that times the inversion with the LPIPS(alex) term in the loss')
    ap.add_argument('--no-precision-ab', action='store_true', help="skip the leg that times the same job with precision 'f16s' (3 matrix instructions everywhere)")
    ap.add_argument('--force-launcher', action='store_true',
                    help='start the rank processes from this process even for --gpus 1 (the path `python bench.py --gpus N` takes for N > 1)')
    return ap.parse_args()


def self_launch(a):
    """``python bench.py --gpus N`` with no launcher around it (WORLD_SIZE unset): this process — which has not imported torch
    and has not touched a GPU — starts the N ranks as CHILD processes (never an exec) with the environment torch.distributed.run
    would give them (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), forwards rank 0's
    stdout (the ONE JSON line) unchanged, waits for all of them and exits non-zero if any rank did; a rank that dies takes the
    others down (they would wait in the rendezvous forever) and every failing rank's stderr tail is shown.
    Analogue in the reference: BasicSR/scripts/dist_train.sh:15-16 (its only launcher; inference there is one process)."""
    import subprocess
    import tempfile
    n = a.gpus
    rc = 1
    for attempt in range(2):            # the port is picked by bind(0) and released before the ranks bind it: retry once if another process took it
        rc, rendezvous_failed = _self_launch_once(a, n, subprocess, tempfile)
        if rc == 0 or not rendezvous_failed:
            break
    return rc


def _self_launch_once(a, n, subprocess, tempfile):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    argv = [x for x in sys.argv[1:] if x != '--force-launcher']
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK='0',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OODGAN_BENCH_SELF_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL's cross-process buffers need it on this driver
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
        # rank 0's stderr streams through (progress is visible and survives a killed parent); the other ranks' goes to a temp file shown on failure
        log = None if r == 0 else tempfile.TemporaryFile(mode='w+', prefix=f'bench_rank{r}_')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=os.getcwd(),
                                      stdout=(None if r == 0 else subprocess.DEVNULL), stderr=log))
    rcs = [None] * n
    import signal
    signal.signal(signal.SIGTERM, lambda *_: sys.exit(143))     # a terminated parent takes the except path below instead of orphaning its ranks
    try:
        _wait_ranks(procs, rcs)
    except BaseException:               # the parent is interrupted (SIGINT, SIGTERM via KeyboardInterrupt / SystemExit): its own children go with it
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10.0      # ... and are waited for: a rank inside an RCCL collective or a long kernel may ignore SIGTERM
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()                        # exact PID of our own child
        raise
    return _report_ranks(n, rcs, logs)


def _wait_ranks(procs, rcs):
    n = len(procs)
    failed_at = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if failed_at is None and any(rc not in (None, 0) for rc in rcs):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 10.0:
            for r, p in enumerate(procs):                       # exact PIDs of our own children only
                if rcs[r] is None:
                    p.terminate()
            failed_at = float('inf')
            t_kill = time.monotonic() + 20.0
        if failed_at == float('inf') and time.monotonic() > t_kill:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()
        time.sleep(0.05)


def _report_ranks(n, rcs, logs):
    """-> (exit code, True if the failure looks like a lost race for the rendezvous port)."""
    bad = [r for r, rc in enumerate(rcs) if rc != 0]
    port_lost = False
    for r, log in enumerate(logs):
        txt = ''
        if log is not None:
            log.seek(0)
            txt = log.read()
            log.close()
        if r in bad and ('Address already in use' in txt or 'EADDRINUSE' in txt):
            port_lost = True
        if r in bad:
            sys.stderr.write(f'--- bench.py rank {r} exited with code {rcs[r]}; stderr tail ---\n' + (txt[-3000:] if log is not None else '(streamed above)') + ('' if txt.endswith('\n') else '\n'))
    if bad:
        sys.stderr.write(f'bench.py: {len(bad)} of {n} rank processes failed (ranks {bad})\n')
        return 1, port_lost
    return 0, False


class ConvProbe:
    """HIP-event timing of every launch of the dominant kernel inside the timed region.  Dominant kernel (largest
    share of GPU time in profiles/): the plain 3x3 stride-1 implicit-GEMM conv with S-form input and >= 64 input
    channels — template instances ``conv_f16s_s1big_kernel<false, false>`` (forward) and ``<true, *>`` (input gradient + style-gradient dot) (split-f16) / ``conv_mfma_kernel<0, 2>`` (fp32):
    forward of the plain ModulatedConv2d layers, their input gradients, and the dense AlignNet convs.
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self, ops):
        self.ops, self.orig, self.recs, self.on = ops, ops.conv3x3, [], False
        self.recs_x = []        # the in-loop 1024² ModulatedConv2d (conv_f16s_stripx forward: fp32 F-form in / out): (e0, e1, bytes, flops)

    def install(self):
        probe = self

        def conv3x3(x, wpk, M, mode=0, **kw):
            B, K, H, W = x.shape
            # work items (16x32-pixel tiles x 64-channel blocks) of the dominant kernel; layers with fewer than 256 of them
            # (4x4 ... 32x32 images) are dispatched to the latency-oriented instance conv_f16s_s1v2_kernel<1, 1, 2>
            items = ((H + 15) // 16) * ((W + 31) // 32) * B * ((M + 63) // 64)
            if (probe.on and isinstance(x, probe.ops.FForm) and kw.get('xf_act') is None and mode == probe.ops.CONV_S1
                    and not torch.cuda.is_current_stream_capturing()):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = probe.orig(x, wpk, M, mode, **kw)
                e1.record()
                # s (B Ci H² + B Co H² + Co Ci 9) + s B (512 + Ci), s = 4 (+ the 3-channel ToRGB sums it also writes)
                probe.recs_x.append((e0, e1, 4.0 * (B * K * H * W + B * M * H * W + K * M * 9) + 4.0 * B * (512 + K) + (12.0 * B * H * W if kw.get('rgb') is not None else 0.0),
                                     2.0 * B * K * M * 9 * H * W))
                return r
            if torch.cuda.is_current_stream_capturing() or not (probe.on and mode == probe.ops.CONV_S1 and M >= 64 and K >= 64 and
                    (wpk.precision == 'f32' or (isinstance(x, probe.ops.SForm) and items >= 256))):
                return probe.orig(x, wpk, M, mode, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = probe.orig(x, wpk, M, mode, **kw)
            e1.record()
            # instance: since round 4 the forward launches in front of a ToRGB also write the ToRGB partial sums and the next up-conv's
            # S-form (oodgan_conv_args.ys / rgb_y: the work of the torgb_fwd_sform pass, +4 bytes per output element)
            fused = kw.get('ys') is not None
            tag = 'input gradient + style-gradient dot' if kw.get('dotx') is not None else (
                'forward + ToRGB sums + S-form of the next conv (fused outputs)' if fused else 'forward')
            # outputs the launch writes: the fp32 tensor (unless want_y=False: inside the W+ loop the fused launches keep the activation only
            # as the S-form, engine.save_sform_only) and the S-form of the next conv (4 bytes per element as well); + dotx for the input gradient
            n_out = (0 if kw.get('want_y') is False else 1) + (1 if fused else 0)
            n_in = 2 if kw.get('dotx') is not None else 1
            probe.recs.append((e0, e1, 2.0 * B * K * M * 9 * H * W,
                               4.0 * (B * K * H * W + (n_in - 1) * B * M * H * W + B * M * H * W * n_out + K * M * 9), tag))
            return r

        self.ops.conv3x3 = conv3x3
        import oodgan.engine as eng
        eng.ops.conv3x3 = conv3x3

    def summary_inloop(self):
        if not self.recs_x:
            return None
        ms = [a.elapsed_time(b) for a, b, _, _ in self.recs_x]
        t, by, fl = sum(ms) / len(ms), self.recs_x[0][2], self.recs_x[0][3]
        return dict(workload='the fp32-I/O ModulatedConv2d 3x3 32->32 @1024x1024 INSIDE the W+ loop (last styled conv of the generator, forward: in-kernel '
                             'conversion of the F-form input, demodulation, noise, bias, lrelu, ToRGB sums)', kernel='conv_f16s_stripx_kernel<false, true, ...>',
                    bound='hbm', achieved=round(by / t / 1e6, 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(by / t / 1e6 / HBM_PEAK_GBPS, 4), ms=round(t, 4),
                    alg_bytes=by, tflops=round(fl / t / 1e9, 1), launches=len(ms), median_ms=round(sorted(ms)[len(ms) // 2], 4), max_ms=round(max(ms), 4),
                    measured_on='exclusive single-stream pass, HIP events around each launch (frac from the MEAN duration)')

    def summary(self):
        if not self.recs:
            return None
        times = [a.elapsed_time(b) for a, b, _, _, _ in self.recs]
        ms = sum(times)
        flops = sum(f for _, _, f, _, _ in self.recs)
        byts = sum(g for _, _, _, g, _ in self.recs)
        n = len(self.recs)
        inst = {}
        for t, (_, _, f, g, tag) in zip(times, self.recs):
            d = inst.setdefault(tag, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += t; d[2] += f; d[3] += g
        instances = {k: dict(launches=v[0], avg_ms=round(v[1] / v[0], 4), tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                             alg_bytes_per_launch=round(v[3] / v[0])) for k, v in inst.items()}
        return dict(launches=n, avg_ms=ms / n, tflops=flops / (ms * 1e-3) / 1e12, flops_per_launch=flops / n,
                    bytes_per_launch=byts / n, instances=instances, median_ms=sorted(times)[n // 2], max_ms=max(times))


def pmc_traffic_instances(a):
    """The same PMC measurement per template instance (forward / input gradient) with each instance's own algorithmic bytes:
    the blended `traffic` over the blended `alg_bytes_per_launch` overstates the re-fetch (the input gradient also reads dotx)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            return json.load(f).get(f'{a.precision}_b{a.batch}_s{a.size}', {}).get('instances')
    except (OSError, ValueError):
        return None


def pmc_step_traffic(a):
    """HBM bytes of one steady-state W+ step from the committed PMC passes (tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE over `tools/wplus_only.py` at two step counts, difference / step difference); None when not on file."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            rec = json.load(f)
            base = f'wplus_step_{a.precision}_b{a.batch}_s{a.size}'
            return rec.get(f'{base}_streams{a.streams}') or rec.get(base)         # the pass recorded at this run's stream count, if there is one
    except (OSError, ValueError):
        return None


def pmc_traffic(a):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    MI355X_MICROARCH.md §HBM), measured on this same command and committed under profiles/ (a PMC pass cannot run
    inside the timed bench); None when no measurement for this configuration is on file."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            rec = json.load(f)
        key = f'{a.precision}_b{a.batch}_s{a.size}'
        return rec.get(key, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def wplus_step_algorithmic(B, size, channel_multiplier=2, breakdown=False, hi_records=False):
    """Algorithmic HBM bytes and flops of ONE steady-state W+ step (forward + backward to W+ + Adam) as the loop is scheduled
    (DESIGN.md §4-5): every tensor a launch must read or write counted once per launch, at 4 bytes per element (fp32, the 64-byte
    S-form / F-form records of 16 channels, and the phase-split S-form all hold 4 bytes per element); weights, noise maps, per-channel
    vectors and partial sums are left out (< 1 %).  flops = 2*B*K*M*9*H*W per styled conv forward and once more for its input gradient
    (no weight gradients exist in this loop), + the ToRGB contractions; the folded-blur kernels' doubled taps are NOT counted.
    ``breakdown``: also the bytes per role of the step (tools/step_ledger.py sets them against the PMC bytes of the kernels playing the role).
    ``hi_records`` (precision 'f16s-g2'): the gradient tensors between the blur^T strip producer, the stride-2 conv, its fused epilogue and the stride-1
    input-gradient conv are 32-byte hi-only records — 2 bytes per element instead of 4 — from the 64² level up (below it and at the 1024² level's
    F-form conv nothing changes)."""
    from oodgan.synth import generator_channels
    ch = generator_channels(channel_multiplier)
    e = 4.0 * B
    roles = dict.fromkeys(('up-conv (transposed / one-pass)', 'up-conv tail (blur + noise + bias + act)', 'conv forward (+ ToRGB, next S-form)',
                           'conv input gradient (+ dot)', 'blur^T + phase split', 'stride-2 conv + fused activation backward', 'MSE + skip pyramid'), 0.0)
    fl = 0.0
    res = 4
    # 4x4: conv1 forward (in + out), its ToRGB, backward (g in, dotx, dx out)
    roles['conv forward (+ ToRGB, next S-form)'] += e * ch[4] * 16 * 3
    roles['conv input gradient (+ dot)'] += e * ch[4] * 16 * 3
    fl += 2 * 2.0 * B * ch[4] * ch[4] * 9 * 16
    cin = ch[4]
    while res < size:
        res *= 2
        C, r2, q2, z2 = ch[res], res * res, (res // 2) ** 2, (res + 1) ** 2
        last = res == size
        # ---- forward
        if last and C <= 32:
            roles['up-conv (transposed / one-pass)'] += e * (cin * q2 + C * r2)      # one-pass up-conv (conv_f16s_upvb): S-form in, F-form out
        else:
            roles['up-conv (transposed / one-pass)'] += e * (cin * q2 + C * z2)      # transposed conv: S-form in, (2H+1)² result out
            roles['up-conv tail (blur + noise + bias + act)'] += e * (C * z2 + 2 * C * r2)     # result in, fp32 planes + next conv's S-form out
        # conv: S-form (F-form) in; out = S-form of the next up-conv only (64² ... 512²), F-form (last), fp32 (low res)
        roles['conv forward (+ ToRGB, next S-form)'] += e * (C * r2 + C * r2)
        if res < 64:
            roles['conv forward (+ ToRGB, next S-form)'] += e * (C * r2 + (0 if last else C * r2))   # low resolution: separate ToRGB pass
        fl += 2.0 * B * 9 * (cin * C * q2 + C * C * r2) + 2.0 * B * 3 * C * r2
        # ---- backward
        hg = 0.5 if (hi_records and res >= 64) else 1.0                             # gradient records of this level's up-conv
        hc = 0.5 if (hi_records and 64 <= res < size) else 1.0                      # ... of this level's conv layer (S-form input of its input gradient)
        hp = 0.5 if (hi_records and 128 <= res) else 1.0                            # ... written by this level's stride-2 conv for the conv layer BELOW (64² and up)
        roles['conv input gradient (+ dot)'] += e * (2 + hc) * C * r2               # gradient S-form (F-form: its own output) in, dotx in, dx out
        roles['blur^T + phase split'] += e * (C * r2 + hg * C * z2)                  # pre-activation gradient in, phase-split S-form out
        roles['stride-2 conv + fused activation backward'] += e * (hg * C * z2 + (1 + hp) * cin * q2)   # phases in, saved activation in, S-form gradient out
        fl += 2.0 * B * 9 * (cin * C * q2 + C * C * r2) + 2.0 * B * 3 * C * r2
        cin = C
    roles['MSE + skip pyramid'] += e * 3 * size * size * 4                         # image + target in, gradient out; skip pyramid down
    by = sum(roles.values())
    return (by, fl, roles) if breakdown else (by, fl)


def modconv_roofline(iters=30, warmup=3):
    """BASELINE.json's second metric ("modulated-conv2d GB/s", configs[4]): ModulatedConv2d.forward (model.py:233-274) of the
    1024² layer in fp16 — x (16,32,1024,1024) f16, 32 -> 32 channels, + noise + bias + leaky ReLU.  ONE timed iteration =
    the whole op as the reference defines it: style affine (EqualLinear 512 -> 32 on the latent) + weight modulation +
    demodulation + f16 packing (one launch since round 3), and the 3x3 conv kernel; algorithmic bytes per SURVEY.md §8(d)
    (2.147 GB, the affine's and the weights' bytes included) over the HIP-event time of that sequence, against the 8 TB/s
    HBM peak; the iterations are pipelined over two HIP streams (see below), the single-stream figure is reported beside it.  The activations live in the f16 channel-blocked layout of csrc/conv_f16.hip (H-form: what an fp16 pipeline
    of this layer would keep between layers); `kernel_ms` is the conv kernel alone.  Runs after the timed region."""
    from oodgan import ops
    B, C, H = 16, 32, 1024
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device='cpu').manual_seed(0)
    xh = ops.HForm(B, C, H, H, dev)
    per = xh.buf.numel() // B

    class _Sample:                               # H-form view of one sample (fill without a (16,32,1024,1024) fp32 tensor)
        def __init__(self, b):
            self.buf, self.B, self.C, self.H, self.W = xh.buf[b * per:(b + 1) * per], 1, C, H, H

        def data_ptr(self):
            return self.buf.data_ptr()

    for b in range(B):
        ops.to_hform(torch.randn(1, C, H, H, generator=g).to(dev), out=_Sample(b))
    wgt = torch.randn(1, C, C, 3, 3, generator=g).to(dev)
    lat = torch.randn(B, 512, generator=g).to(dev)
    mod_w = torch.randn(C, 512, generator=g).to(dev)
    mod_b = torch.ones(C, device=dev)
    noise = torch.randn(B, 1, H, H, generator=g).to(dev)
    bias = (0.1 * torch.randn(C, generator=g)).to(dev)
    nw = torch.tensor([0.1], device=dev)
    out = ops.HForm(B, C, H, H, dev)

    wbuf = [None]

    def op():
        # EqualLinear(512, Ci, bias_init=1) (model.py:223,236) + scale*W*s + demodulate (model.py:237-241): ONE launch
        # (oodgan_modconv_f16_pack_affine), into a reused weight buffer
        packed = ops.modconv_f16_pack(wgt, None, act='lrelu', latent=lat, mod_weight=mod_w, mod_bias=mod_b, out=wbuf[0])
        wbuf[0] = packed[0]
        ops.modconv_f16(xh, packed, noise, nw, bias, out=out)        # grouped conv + noise + bias + act model.py:268-272,343-350
        return packed

    for _ in range(warmup):
        packed = op()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op()
    e1.record()
    torch.cuda.synchronize()
    ms_serial = e0.elapsed_time(e1) / iters

    # SECONDARY figures (named as such in the line, never the headline): the same sequence as a two-stage pipeline over the
    # iterations (independent samples of the op) — the weight preparation of iteration i+1 (one small launch) on a second HIP
    # stream under the conv of iteration i, into the other of two weight buffers — launched eagerly and replayed from a hipGraph.
    side = torch.cuda.Stream(device=dev)
    wb2 = [None, None]
    ev_pack = [torch.cuda.Event(), torch.cuda.Event()]
    ev_conv = [torch.cuda.Event(), torch.cuda.Event()]

    def pack_async(i):
        with torch.cuda.stream(side):
            if i >= 2:
                side.wait_event(ev_conv[i & 1])      # the conv that last read this weight buffer (iteration i-2)
            pk = ops.modconv_f16_pack(wgt, None, act='lrelu', latent=lat, mod_weight=mod_w, mod_bias=mod_b, out=wb2[i & 1])
            wb2[i & 1] = pk[0]
            ev_pack[i & 1].record(side)
        return pk

    def pipelined(n):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        nxt = pack_async(0)
        for i in range(n):
            pk = nxt
            if i + 1 < n:
                nxt = pack_async(i + 1)
            cur.wait_event(ev_pack[i & 1])
            ops.modconv_f16(xh, pk, noise, nw, bias, out=out)
            ev_conv[i & 1].record(cur)
        cur.wait_stream(side)

    pipelined(warmup + 2)
    torch.cuda.synchronize()
    e0.record()
    pipelined(iters)
    e1.record()
    torch.cuda.synchronize()
    ms_eager = e0.elapsed_time(e1) / iters
    ms_graph = None
    try:
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=cap):
            pipelined(iters)
        torch.cuda.current_stream().wait_stream(cap)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms_graph = e0.elapsed_time(e1) / iters
    except Exception as ex:
        sys.stderr.write('modconv2d: graph replay of the pipelined sequence not available (%s)\n' % (str(ex).splitlines()[0] if str(ex) else type(ex).__name__))
    e0.record()
    for _ in range(iters):
        ops.modconv_f16(xh, packed, noise, nw, bias, out=out)
    e1.record()
    torch.cuda.synchronize()
    kms = e0.elapsed_time(e1) / iters
    alg = 2 * (B * C * H * H + B * C * H * H + C * C * 9) + 2 * B * (512 + C)
    flops = 2.0 * B * C * C * 9 * H * H
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            traffic = json.load(f).get('modconv_f16_b16_s1024', {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        pass
    gbps = alg / ms_serial / 1e6
    frac_of = lambda t: None if t is None else round(alg / t / 1e6 / HBM_PEAK_GBPS, 4)
    return dict(workload='fp16 ModulatedConv2d 3x3 32->32 @1024x1024, batch 16: style affine + modulate/demodulate/pack + conv (+noise, bias, lrelu)',
                kernel='modconv_f16_strip_kernel', bound='hbm', achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                frac=round(gbps / HBM_PEAK_GBPS, 4), ms=round(ms_serial, 4), kernel_ms=round(kms, 4), frac_kernel_alone=frac_of(kms),
                alg_bytes=alg, traffic=traffic, tflops=round(flops / ms_serial / 1e9, 1),
                ms_pipelined_eager=round(ms_eager, 4), frac_pipelined_eager=frac_of(ms_eager),
                ms_pipelined_graph=None if ms_graph is None else round(ms_graph, 4), frac_pipelined_graph=frac_of(ms_graph),
                note='ms / achieved / frac: ONE ModulatedConv2d.forward at a time — the two launches (weight preparation, conv) back to back '
                     'on one stream, HIP events around the iterations (the BASELINE definition; rounds 1-2 reported this, round 3 '
                     'reported the pipelined figure as `frac` and this one as `frac_single_stream`).  ms_pipelined_*: secondary — the '
                     'weight preparation of iteration i+1 on a second stream under the conv of iteration i, eager / hipGraph replay')


def generator_fwd_b4(a, dev, reps=9):
    """BASELINE configs[1]: StyleGAN2 1024² generator forward, batch 4 — the reference call
    ``Generator([z], noise=<list>)`` (model.py:483-585: mapping MLP, 17 styled convs, 9 ToRGB) on recipe weights, inputs of
    tests/make_golden_params.py (the parity run of this exact workload is tests/test_hip_generator.py::
    test_generator_1024_batch4_vs_reference).  Latency = median host wall time incl. synchronize; GB/s on the algorithmic bytes of
    SURVEY.md Appendix B (Σ(W + in + out) over the 26 convs = 1 291 MB per image in fp32, 148.5 GFLOP)."""
    import statistics
    from oodgan import ops, synth
    from oodgan.modules import Generator
    size, B = a.size, 4
    out = {}
    z = synth.normal('gen_b4.z', (B, 512), 21).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=2100)]
    sd = synth.generator_state(size, seed=0)
    saved = ops.PRECISION
    try:
        for prec in ('f16s', 'f32'):
            ops.PRECISION = prec
            G = Generator(size, 512, 8)
            G.load_state_dict(sd, strict=True)
            G = G.to(dev).eval()
            for _ in range(2):
                G([z], noise=noises)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                img, _ = G([z], noise=noises)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = statistics.median(ts)
            out[prec] = dict(latency_ms=round(t * 1e3, 3), images_per_s=round(B / t, 2), gbps=round(B * 1291e6 / t / 1e9, 1),
                             frac_hbm=round(B * 1291e6 / t / 1e9 / HBM_PEAK_GBPS, 4), tflops=round(B * 148.5e9 / t / 1e12, 1))
            del G, img
    finally:
        ops.PRECISION = saved
    out.update(batch=B, alg_bytes_per_image=1291e6, alg_flops_per_image=148.5e9,
               note=f'Generator([z], noise=list) at {size}x{size}, batch {B}: median of {reps} host-timed calls incl. synchronize; f16s = the '
                    'split-f16 matrix-core arithmetic (fp32-class, default), f32 = the exact-fp32 MFMA variant')
    return out


def cpu_baseline(size):
    """Reported baseline: the CPU oracle (same ATen graph as the reference's PyTorch path) on this box's
    host threads, B = 1: one W+ step (forward + backward + Adam) timed as 1 warm-up + median of 3, and the final OOD
    forward (post-encoder: generator + SAMM 2 cycles x 4 levels + mask blend) timed once after it; one inversion =
    100 steps + 1 OOD forward."""
    import statistics
    from oracle import ref_cpu as R
    from oodgan import synth
    n = torch.get_num_threads()
    P = synth.ood_state(size, seed=0)
    G = {k[len('generator.'):]: v for k, v in P.items() if k.startswith('generator.')}
    lat = synth.make_latents(size, 1, seed=3000, std=0.3)
    noises = synth.make_noises(size, 1, seed=2000)
    target = synth.make_images(size, 1, seed=1000)
    ts = []
    for _ in range(4):
        t0 = time.time()
        R.wplus_invert(G, target, lat, noises, size, steps=1)
        ts.append(time.time() - t0)
    step = statistics.median(ts[1:])
    t0 = time.time()
    with torch.no_grad():
        R.ood_forward(P, target, lat, synth.make_encoder_feats(1, seed=4000), noises, size)
    ood = time.time() - t0
    return dict(value=1.0 / (100.0 * step + ood), unit='images/s', cores=n, kind='port', batch=1, wplus_step_s=round(step, 3), ood_forward_s=round(ood, 3),
                sample=f'B=1 at {size}x{size} on {n} threads: 1 W+ step (fwd+bwd+Adam) = {step:.2f}s (median of 3 after 1 warm-up: '
                       f'{", ".join(f"{t:.2f}" for t in ts)}), final OOD forward = {ood:.2f}s (1 run); inversion = 100 steps + 1 OOD forward')


def build_full_model(a, dev):
    """``ood_faceGAN_e4e`` WITH its e4e encoder (recipe weights), as the CLI builds it."""
    from oodgan import synth
    from oodgan.arch import ood_faceGAN_e4e
    m = ood_faceGAN_e4e(out_size=a.size, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08, cycle_align=2,
                        blend_with_gen=True, ModSize=256)
    sd = synth.ood_state(a.size, seed=0)
    enc = synth.encoder_state({k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}, seed=41)
    sd.update({'encoder.' + k: (v * 0.1 if k.endswith('linear.weight') else v) for k, v in enc.items()})
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()


def end_to_end(a, m, x, noises):
    """The same inversion INCLUDING the step before the path: the e4e encoder (IR-SE-50 + FPN + 18
    GradualStyleBlocks, `Encoder4EditingHIP` on the HIP conv kernels, recipe weights) predicts the start latents and the
    4-level feature pyramid from the 256x256-pooled input; then the timed workload of `value` runs unchanged.
    1 warm-up + 1 timed inversion of the same batch; reported beside `value`."""
    B = x.shape[0]
    run = lambda: m.invert(x, steps=a.wsteps, noise=noises, streams=a.streams)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.encode(x)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, losses = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=round(B / dt, 4), unit='images/s', ms_per_step=round(dt * 1e3, 2), encoder_ms=round(t_enc * 1e3, 2),
                final_loss_mean=float(losses[-1].mean().item()),
                note=f'encoder (Encoder4EditingHIP, batch {B} at 256x256) + {a.wsteps} W+ steps + OOD forward; recipe encoder weights')


def forward_only(a, m, x, noises, reps=7, only_full=False):
    """The reference's OWN hot path: ``model(input_im)`` — e4e encoder at 256² + OOD forward (generator
    with the four SAMM hooks, mask compose, blend), no W+ steps — the call run_ood_faceGAN_inversion.py:167-172 brackets with
    ``time.time()`` + ``torch.cuda.synchronize()`` ("Average process time", :187).  B=1 latency as that script runs it
    (median of ``reps`` after 2 warm-ups) and B=8 throughput; the legs (encoder / OOD forward) from HIP events."""
    import statistics
    B = x.shape[0]
    out = {}
    for tag, b in ((f'b{B}', B),) if only_full else (('b1', 1), (f'b{B}', B)):
        xb = x[:b].contiguous()
        nb = [n[:b].contiguous() for n in noises]
        for _ in range(2):
            m(xb, noise=nb)
        torch.cuda.synchronize()
        ts, enc_ms, ood_ms = [], [], []
        for _ in range(reps):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t0 = time.perf_counter()
            e0.record()
            lats, feats = m.encode(xb)
            e1.record()
            m._ood_forward(xb, lats, feats, noise=nb)
            e2.record()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            enc_ms.append(e0.elapsed_time(e1))
            ood_ms.append(e1.elapsed_time(e2))
        t = statistics.median(ts)
        out[tag] = dict(batch=b, latency_ms=round(t * 1e3, 3), images_per_s=round(b / t, 3),
                        encoder_ms=round(statistics.median(enc_ms), 3), ood_forward_ms=round(statistics.median(ood_ms), 3))
        # the same call replayed from a captured hipGraph (oodgan.arch.GraphedForward): no host launch overhead, no gaps
        from oodgan.arch import GraphedForward
        gf = GraphedForward(m)
        gf(xb, noise=nb)
        torch.cuda.synchronize()
        tg = []
        for _ in range(reps):
            t0 = time.perf_counter()
            gf(xb, noise=nb)
            torch.cuda.synchronize()
            tg.append(time.perf_counter() - t0)
        tgm = statistics.median(tg)
        out[tag]['graph_replay_latency_ms'] = round(tgm * 1e3, 3)
        out[tag]['graph_replay_images_per_s'] = round(b / tgm, 3)
        gf.reset()
        del gf
        # opt-in: forward range scales carried from call to call (OODGAN_CARRY_FORWARD=1 — results then depend on the previous call at
        # fp32-rounding level, which is why it is not the default)
        from oodgan import modules as _mod
        old = _mod.CARRY_FORWARD
        try:
            _mod.CARRY_FORWARD = True
            for _ in range(3):
                m(xb, noise=nb)
            torch.cuda.synchronize()
            tc = []
            for _ in range(reps):
                t0 = time.perf_counter()
                m(xb, noise=nb)
                torch.cuda.synchronize()
                tc.append(time.perf_counter() - t0)
            out[tag]['carried_scales_latency_ms'] = round(statistics.median(tc) * 1e3, 3)
        finally:
            _mod.CARRY_FORWARD = old
    out['note'] = ('model(x): e4e encoder (256x256) + OOD forward (generator + SAMM 2 cycles x 4 levels + mask blend) at '
                   f'{a.size}x{a.size}, host wall time incl. synchronize, median of {reps}; the reference times exactly this call.  Default = every call measures its '
                   'range scales (a pure function of its inputs); carried_scales_latency_ms = the opt-in OODGAN_CARRY_FORWARD=1')
    return out


def main():
    a = parse()
    t_proc = time.perf_counter()
    if 'WORLD_SIZE' not in os.environ and (a.gpus > 1 or a.force_launcher):
        sys.exit(self_launch(a))                # before torch is imported: the parent never initialises a GPU
    global torch
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert a.gpus == world, (f'--gpus {a.gpus} but WORLD_SIZE={world}: under a launcher --gpus must equal the number of ranks it starts '
                             '(without a launcher, `python bench.py --gpus N` starts the N ranks itself)')
    # launched by torch.distributed.run (also with ONE rank): the process group is RCCL ('nccl') and the barrier / MAX
    # all_reduce / all_gather below run through it; a plain `python bench.py` has no process group
    dist_on = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    cpus = None
    if world > 1:
        # one rank per GPU: pin the launch thread near its GPU before anything touches the device (plain sched_setaffinity)
        from oodgan.parallel import bind_rank_to_cpus
        cpus = bind_rank_to_cpus(local_rank)
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU (the hot path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', device_id=dev)

    if a.no_plan:
        os.environ['OODGAN_USE_PLAN'] = '0'
    from oodgan import ops, synth
    ops.PRECISION = a.precision
    from oodgan.arch import ood_faceGAN_e4e
    from oodgan.parallel import shard_slice, gather_latents

    size, B = a.size, a.batch
    model = ood_faceGAN_e4e(out_size=size, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08,
                            cycle_align=2, blend_with_gen=True, ModSize=256, build_encoder=False)
    model.load_state_dict(synth.ood_state(size, seed=0), strict=True)
    model = model.to(dev).eval()
    # per-rank shard of the global synthetic batch (global batch = B*world, contiguous slices by rank); every image,
    # latent, feature pyramid and noise set is generated from its GLOBAL index, so a rank only materialises its own
    # slice and the data of image g does not depend on the number of ranks
    gB = B * world
    sl = shard_slice(gB, rank, world)
    gidx = list(range(sl.start, sl.stop))
    cat = lambda parts: torch.cat(parts, 0).to(dev)
    x = cat([synth.make_images(size, 1, seed=1000 + g) for g in gidx])
    enc_lats = cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in gidx])
    feats_per = [synth.make_encoder_feats(1, seed=4000 + g) for g in gidx]
    enc_feats = [cat([f[i] for f in feats_per]) for i in range(4)]
    noise_per = [synth.make_noises(size, 1, seed=2000 + g) for g in gidx]
    noises = [cat([n[i] for n in noise_per]) for i in range(len(noise_per[0]))]
    del feats_per, noise_per
    torch.cuda.synchronize()

    probe = ConvProbe(ops)
    if not a.no_roofline_events:
        probe.install()

    def one_step():
        out, lats, losses = model.invert(x, steps=a.wsteps, noise=noises, streams=a.streams, use_graph=bool(a.graph), enc_lats=enc_lats, enc_feats=enc_feats)
        return gather_latents(lats, gB) if dist_on else lats, losses

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_proc      # input synthesis + model build + warm-up of THIS rank, before the first barrier
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    probe.on = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        all_lats, losses = one_step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    probe.on = False
    plan_snapshot = (model.last_invert_plan, model.last_invert_stats)       # of the LAST timed inversion (later legs overwrite them)
    per_rank = None
    if dist_on:
        mine = torch.tensor([dt, float(len(cpus) if cpus else 0), t_setup, float(torch.get_num_threads())], device=dev, dtype=torch.float64)
        tmax = mine[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if world > 1:                                   # launch skew / a slow rank must be visible in the line
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = dict(seconds=[round(float(t[0].item()), 4) for t in allr], cpus_bound=[int(t[1].item()) for t in allr],
                            setup_seconds=[round(float(t[2].item()), 2) for t in allr], torch_threads=[int(t[3].item()) for t in allr])
        dt = tmax.item()
    def roofline_of(ps, where):
        if not ps:
            return None
        f16s = a.precision in ('f16s', 'f16s-g2')
        peak = MFMA_F16_PEAK_TFLOPS if f16s else MFMA_F32_PEAK_TFLOPS
        return dict(bound='mfma', achieved=round(ps['tflops'], 3), peak=peak, unit='TFLOP/s',
                    frac=round(ps['tflops'] / peak, 4), traffic=pmc_traffic(a), traffic_per_instance=pmc_traffic_instances(a),
                    kernel=('conv_f16s_s1big_kernel<false|true, *>' if f16s else 'conv_mfma_kernel<0, 2>') +
                           ' (plain 3x3 stride-1 implicit GEMM, >=64 input channels: forward + input gradient)',
                    measured_on=where,
                    alg_bytes_per_launch=ps['bytes_per_launch'],
                    note=(('algorithmic flops; the split-f16 scheme issues 3 MFMAs per product in the forward instances (their ceiling: peak/3 = 833 TFLOP/s)'
                           + (" and 2 in the input-gradient instances with precision 'f16s-g2' (x_hi * (w_hi + w_lo); ceiling peak/2 = 1250 TFLOP/s; 9 instead "
                              'of 14 matrix instructions per 16x16 tile and 16-channel chunk)' if a.precision == 'f16s-g2' else ', forward and input gradient alike'))
                          if f16s else 'exact fp32 MFMA'),
                    launches=ps['launches'], avg_launch_ms=round(ps['avg_ms'], 4), median_launch_ms=round(ps['median_ms'], 4), max_launch_ms=round(ps['max_ms'], 4),
                    alg_flops_per_launch=ps['flops_per_launch'], instances=ps.get('instances'),
                    instances_note=('achieved / frac average ALL launches of the kernel.  Since round 4 the forward launches in front of a ToRGB '
                                    '(three of the eight per W+ step) also produce the ToRGB partial sums and the S-form input of the next up-conv in '
                                    'their epilogue — the work of a separate HBM-bound pass (torgb_fwd_sform, 0.36 ms per step) that no longer '
                                    'runs: the step is 0.1 ms shorter, these launches are ~25 % longer (LABNOTES.md 13.9)') if f16s else None)

    if rank == 0:
        roof_timed = roofline_of(probe.summary(), f'timed region ({a.streams} concurrent HIP streams: launches share the GPU)'
                                 + ('' if a.no_plan else '; with launch plans only the two Python-driven steps of every inversion pass the event wrapper'))
        roof = roof_timed
        inloop = probe.summary_inloop()
        probe.recs_x = []
        if a.streams > 1 and not a.no_roofline_events:
            # A launch duration is a property of the kernel only while the kernel owns the GPU.  The timed region advances
            # sub-batches on concurrent streams (matrix kernels of one beside the HBM-bound producers of the other), so its
            # per-launch durations are inflated by sharing; the roofline of the dominant kernel is therefore taken from an
            # exclusive pass of the SAME workload (full batch, one stream) run here, right after the timed region — the
            # configuration the rocprofv3 kernel statistics under profiles/ are recorded in.
            from oodgan.engine import WPlusInverter
            lats0, _ = model.encode(x, enc_lats=enc_lats, enc_feats=enc_feats)
            # untimed first: this is the process's first full-batch single-stream run — its buffers (1 GB tensors) come from the driver, and a
            # host stall inside a wrapped call would sit between that launch's two events (seen once: 52 ms inside one 0.6 ms launch)
            probe.on = False
            WPlusInverter(model.generator.engine(), use_plan=False).invert(x, lats0, noises, steps=3, streams=1)
            torch.cuda.synchronize()
            probe.recs, probe.recs_x, probe.on = [], [], True
            # launch by launch from Python (use_plan=False): the events sit around every launch of the kernel
            WPlusInverter(model.generator.engine(), use_plan=False).invert(x, lats0, noises, steps=a.roofline_steps, streams=1)
            torch.cuda.synchronize()
            probe.on = False
            roof = roofline_of(probe.summary(), f'exclusive single-stream pass of the same workload (batch {B}, {a.roofline_steps} W+ steps) '
                                                'inside bench.py right after the timed region')
            inloop = probe.summary_inloop()
        # ---- the STEP against both roofs (north_star: "throughput ... as fraction of the HBM roofline"): wall time of one W+ step of the
        # timed region = (inversion - final OOD forward) / steps, the OOD forward timed alone right here (HIP events, same inputs)
        step_roof = None
        if not a.no_roofline_events:
            model._ood_forward(x, all_lats[:B] if all_lats.shape[0] >= B else enc_lats, enc_feats, noise=noises)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model._ood_forward(x, enc_lats, enc_feats, noise=noises)
            e1.record()
            torch.cuda.synchronize()
            ood_ms = e0.elapsed_time(e1)
            step_ms = (dt / a.steps * 1e3 - ood_ms) / max(a.wsteps, 1)
            alg_b, alg_f = wplus_step_algorithmic(B, size, hi_records=(a.precision == 'f16s-g2'))
            pmc = pmc_step_traffic(a)
            # the PMC bytes belong to ONE configuration (precision, batch, stream count: the sub-batch size selects kernels at the 32² / 64² layers);
            # they are set against this run's wall time only when it is the same one
            if pmc and int(pmc.get('streams', 1)) != a.streams:
                pmc = None
            hbm_b = pmc.get('hbm_bytes_per_step') if pmc else None
            step_roof = dict(
                wall_ms_per_wplus_step=round(step_ms, 4), ood_forward_ms=round(ood_ms, 3), streams=a.streams,
                algorithmic_gb_per_step=round(alg_b / 1e9, 3), algorithmic_tflop_per_step=round(alg_f / 1e12, 4),
                hbm_gb_per_step_pmc=None if hbm_b is None else round(hbm_b / 1e9, 3),
                traffic_over_algorithmic=None if hbm_b is None else round(hbm_b / alg_b, 3),
                achieved_tbps_algorithmic=round(alg_b / step_ms / 1e9, 3), frac_hbm_algorithmic=round(alg_b / step_ms / 1e6 / HBM_PEAK_GBPS, 4),
                achieved_tbps_pmc=None if hbm_b is None else round(hbm_b / step_ms / 1e9, 3),
                frac_hbm_pmc=None if hbm_b is None else round(hbm_b / step_ms / 1e6 / HBM_PEAK_GBPS, 4),
                achieved_tflops=round(alg_f / step_ms / 1e9, 1), frac_mfma_f16_peak=round(alg_f / step_ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4),
                frac_of_issued_mfma_ceiling=round((2.5 if a.precision == 'f16s-g2' else 3.0) * alg_f / step_ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4),
                issued_mfma_per_product=('3 forward / 2 input gradient (f16s-g2): 2.5 on average over the step' if a.precision == 'f16s-g2' else 3),
                peak_hbm_gbps=HBM_PEAK_GBPS, peak_mfma_tflops=MFMA_F16_PEAK_TFLOPS,
                pmc_source=None if not pmc else pmc.get('source'),
                note='one W+ step (generator forward + MSE + backward to W+ + Adam, batch %d) of the timed region: wall time = (inversion - final OOD '
                     'forward) / W+ steps; algorithmic bytes / flops: bench.wplus_step_algorithmic (every tensor a launch of the step must read or write, '
                     'once per launch; 2*B*K*M*9*H*W per conv and direction); PMC bytes: 2*FETCH_SIZE + WRITE_SIZE of the step' % B)
        modconv = None
        if not a.no_modconv and world == 1:
            modconv = modconv_roofline()
        single = None
        if not a.no_single_stream and world == 1 and a.streams > 1:
            # the same job on ONE stream (every kernel owns the GPU), reported beside `value`
            ss_run = lambda: model.invert(x, steps=a.wsteps, noise=noises, streams=1, enc_lats=enc_lats, enc_feats=enc_feats)
            ss_run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, _, ml = ss_run()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            single = dict(streams=1, value=round(B / (t2 - t1), 4), unit='images/s', ms_per_step=round((t2 - t1) * 1e3, 2),
                          final_loss_mean=float(ml[-1].mean().item()), note='same workload on one HIP stream')
        def timed_invert(m_, xb, nz, el, ef, **kw):
            m_.invert(xb, steps=a.wsteps, noise=nz, enc_lats=el, enc_feats=ef, **kw)           # warm-up (plans, pools)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, _, ml = m_.invert(xb, steps=a.wsteps, noise=nz, enc_lats=el, enc_feats=ef, **kw)
            torch.cuda.synchronize()
            return time.perf_counter() - t1, ml

        # ---- the launch plan of the timed inversion (oodgan_plan_*): launches per recorded step, steps replayed
        plan_info = None if plan_snapshot[0] is None else dict(
            enabled=not a.no_plan, launches_per_step=plan_snapshot[0]['launches'], steps_replayed=plan_snapshot[0]['steps'],
            steps_run=plan_snapshot[1]['steps_run'], rollbacks=plan_snapshot[1]['rollbacks'],
            note='steps 3..N of every (sub-)batch are one oodgan_plan_run call each (eager launches re-issued from C++); host cost per step measured by '
                 'tools/plan_probe.py: 0.26 ms against 1.5 ms Python-driven (profiles/r6_plan_probe.txt); rollbacks = windows repeated with exact range scales')
        # ---- image 0 of the timed batch against the REFERENCE's 100-step run from the same start latents (tests/golden/make_golden.py:
        # gold_wplus_long('wplus_long_1024_bench0'): reference Generator autograd + torch.optim.Adam, fp32)
        loss_check = None
        fx = os.path.join(ROOT, 'tests', 'golden', 'wplus_long_1024_bench0.npz')
        if rank == 0 and size == 1024 and a.wsteps == 100 and gidx[0] == 0 and os.path.exists(fx):
            import numpy as np
            ref = np.load(fx)['losses'][:, 0]
            if ref.shape[0] == 100:
                mine = losses[:, 0].double().cpu().numpy()
                rel = abs(mine - ref) / ref
                loss_check = dict(image=0, final_loss=float(mine[-1]), reference_final_loss=float(ref[-1]), rel_final=float(rel[-1]),
                                  max_rel_steps_1_20=float(rel[:20].max()), max_rel_all_steps=float(rel.max()),
                                  reference='reference Generator autograd + torch.optim.Adam, 100 steps, fp32 (tests/golden/wplus_long_1024_bench0.npz)')
                assert rel[:20].max() < 1e-3 and rel[-1] < 1e-3, f'W+ loss curve of image 0 left the reference band: {loss_check}'
        b1 = None
        if not a.no_b1 and world == 1:
            # the reference CLI's mode: one image at a time (run_ood_faceGAN_inversion.py:158-182)
            dt1, ml1 = timed_invert(model, x[:1], [n[:1] for n in noises], enc_lats[:1], [f[:1] for f in enc_feats], streams=1)
            b1 = dict(batch=1, value=round(1.0 / dt1, 4), unit='images/s', ms_per_inversion=round(dt1 * 1e3, 2), ms_per_wplus_step=round(dt1 * 1e3 / max(a.wsteps, 1), 3),
                      final_loss=float(ml1[-1].mean().item()), note=f'B = 1: {a.wsteps} W+ steps + OOD forward of ONE image on one stream')
        lp_leg = None
        if not a.no_lpips and world == 1:
            dtl, mll = timed_invert(model, x, noises, enc_lats, enc_feats, streams=a.streams, lpips_weight=0.8)
            terms = model.last_loss_terms
            lp_leg = dict(value=round(B / dtl, 4), unit='images/s', ms_per_step=round(dtl * 1e3, 2), lpips_weight=0.8,
                          first_total_loss_mean=float(mll[0].mean().item()), final_total_loss_mean=float(mll[-1].mean().item()),
                          final_mse_mean=float(terms['mse'][-1].mean().item()), final_lpips_mean=float(terms['lpips'][-1].mean().item()),
                          note='the same inversion with loss = MSE + 0.8 * LPIPS(alex) (north_star: "W+ Adam steps against LPIPS/L2"; oodgan/lpips.py, csrc/lpips.hip: exact-fp32 MFMA '
                               'convs). SEEDED AlexNet / lin weights — the lpips package and its weights are absent (SURVEY §8c): PARITY UNPINNED, timing only; never `value`')
        prec_ab = None
        if not a.no_precision_ab and world == 1 and a.precision == 'f16s-g2':
            ops.PRECISION = 'f16s'
            try:
                m3 = ood_faceGAN_e4e(out_size=size, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08, cycle_align=2, blend_with_gen=True,
                                     ModSize=256, build_encoder=False)
                m3.load_state_dict(synth.ood_state(size, seed=0), strict=True)
                m3 = m3.to(dev).eval()
                dt3, ml3 = timed_invert(m3, x, noises, enc_lats, enc_feats, streams=a.streams)
                prec_ab = dict(precision='f16s', value=round(B / dt3, 4), unit='images/s', ms_per_step=round(dt3 * 1e3, 2), final_loss_mean=float(ml3[-1].mean().item()),
                               note="the same job with 3 matrix instructions per product in the input-gradient convs too (the default until round 5)")
                del m3
            finally:
                ops.PRECISION = a.precision
        gen_b4 = None
        if not a.no_generator_fwd and world == 1 and size == 1024:
            gen_b4 = generator_fwd_b4(a, dev)
        e2e = fwd_only = None
        if not (a.no_end_to_end and a.no_forward_only) and world == 1 and size == 1024:
            full = build_full_model(a, dev)
            if not a.no_end_to_end:
                e2e = end_to_end(a, full, x, noises)
            if not a.no_forward_only:
                fwd_only = forward_only(a, full, x, noises)
            del full
        cpu = None if (a.no_cpu_baseline or world > 1) else cpu_baseline(size)     # rank 0 at N=1 only
        if fwd_only is not None and cpu is not None:
            fwd_only['cpu_baseline'] = dict(ood_forward_s=cpu['ood_forward_s'], cores=cpu['cores'], kind='port', batch=1,
                                            note='CPU oracle, OOD forward after the encoder (the encoder is not part of the oracle)')
        line = {
            'metric': '1024² face inversions/sec (100 W+ steps)', 'value': round(gB * a.steps / dt, 4), 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f16s': 'f16-split (hi+lo f16 operands, 3 MFMA/product, f32 accumulate; fp32-equivalent)',
                      'f16s-g2': 'f16-split (hi+lo f16 operands, f32 accumulate): 3 MFMA/product in every forward conv (fp32-equivalent), 2 in the input-gradient convs '
                                 '(back-propagated gradient rounded to f16 before each contraction, weights keep hi+lo); dL/dW+ vs the float64 reference 1.4e-5..3.6e-5 '
                                 "('f16s': 1.3e-5..2.6e-5, the reference's own fp32: 2.5e-5), 100-step loss curve within 5.5e-5 of the reference's",
                      'f32': 'f32'}[a.precision],
            'data': 'synthetic',
            'config': {'workload': f'OOD inversion loop: {a.wsteps} W+ Adam steps (fixed noise, per-image MSE) + 1 OOD '
                                   f'forward (SAMM 2 cycles x 4 levels, mask blend), {size}x{size}, batch {B} per GPU',
                       'global_batch': gB, 'image_size': size, 'wplus_steps': a.wsteps, 'parallelism': f'batch-shard x{world}',
                       'collective_backend': (dist.get_backend() if dist_on else None), 'launcher': ('bench.py self-launch' if os.environ.get('OODGAN_BENCH_SELF_LAUNCHED') else ('external' if dist_on else None)), 'gathered_latents': list(all_lats.shape), 'streams_per_gpu': a.streams, 'hipgraph_replay': bool(a.graph),
                       'final_loss_mean': float(losses[-1].mean().item()), 'first_loss_mean': float(losses[0].mean().item())},
            'roofline': roof,
            'step_roofline': step_roof,
            'roofline_timed_region': roof_timed if roof is not roof_timed else None,
            'modconv2d': modconv,
            'modconv2d_in_loop': inloop,
            'launch_plan': plan_info,
            'loss_check': loss_check,
            'wplus_b1': b1,
            'wplus_lpips': lp_leg,
            'precision_f16s': prec_ab,
            'single_stream': single,
            'end_to_end': e2e,
            'forward_only': fwd_only,
            'generator_fwd_b4': gen_b4,
            'per_rank': per_rank,
            'cpu_baseline': cpu,
        }
        print(json.dumps(line, ensure_ascii=False))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
