#!/usr/bin/env python3
"""Host cost of one W+ step: the Python-driven step (~170 ctypes calls) against the recorded launch plan
(oodgan_plan_run, include/oodgan.h), and what eight rank processes sharing one host do to it.

    python tools/plan_probe.py [--batch 4] [--size 1024] [--procs 8]

 1. host ms per step, Python-driven: wall time of enqueueing 6 steady-state steps without synchronising (the GPU queue absorbs them);
 2. host ms per step, plan-driven: the same for oodgan_plan_run (real launches);
 3. the plan with the NULL launch backend (oodgan_plan_set_null_launch: closures run, the launch itself is skipped): pure C++ replay cost;
 4. --procs N: N processes, each pinned to its own slice of the host's CPUs (oodgan.parallel.bind_rank_to_cpus — what bench.py's ranks do),
    record their plan one after the other and then replay it with the null backend AT THE SAME TIME; per-process time against (3).
Child mode (internal): --child RANK NPROCS GO_FILE."""
import argparse
import json
import os
import subprocess
import sys
import time

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))


def build(a):
    import torch
    from oodgan import synth
    from oodgan.engine import GeneratorEngine, WPlusInverter, _WRun
    dev = torch.device('cuda:0')
    B, size = a.batch, a.size
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size, precision=a.precision)
    target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
    noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(len(synth.make_noises(size, 1, seed=2000)))]
    w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
    return torch, eng, target, w0, noises, WPlusInverter, _WRun


def make_run(torch, eng, target, w0, noises, WPlusInverter, _WRun, use_plan, steps=1000):
    inv = WPlusInverter(eng, use_plan=use_plan, check_every=0)
    eng.reset_bwd_state()
    eng.reset_fwd_state()
    inv._runs = []
    run = _WRun(inv, eng, target, w0, noises, steps, None, True, False)
    for _ in range(3):              # exact step, recorded step, one replay / one more eager step
        run.advance()
    torch.cuda.synchronize()
    return inv, run


def host_ms(torch, fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) * 1e3 / n
    torch.cuda.synchronize()
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--procs', type=int, default=8)
    ap.add_argument('--precision', default='f16s-g2')
    ap.add_argument('--child', nargs=3, default=None)
    a = ap.parse_args()
    if a.child:
        return child(a)
    parts = build(a)
    torch = parts[0]
    from oodgan import _lib
    res = {'batch': a.batch, 'size': a.size, 'precision': a.precision}
    _, run_e = make_run(*parts, use_plan=False)
    res['python_driven_host_ms_per_step'] = round(min(host_ms(torch, run_e.advance, 6) for _ in range(3)), 3)
    del run_e
    _, run_p = make_run(*parts, use_plan=True)
    assert run_p.plan is not None
    res['plan_launches'] = run_p.plan.size
    res['plan_driven_host_ms_per_step'] = round(min(host_ms(torch, run_p.advance, 6) for _ in range(3)), 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run_p.advance()
    torch.cuda.synchronize()
    res['plan_driven_gpu_ms_per_step'] = round((time.perf_counter() - t0) * 1e3 / 20, 3)
    _lib.lib().oodgan_plan_set_null_launch(1)
    res['plan_null_backend_host_ms_per_step'] = round(min(host_ms(torch, lambda: run_p.plan.run(50), 1) / 50 for _ in range(5)), 4)
    _lib.lib().oodgan_plan_set_null_launch(0)
    del run_p
    print(json.dumps(res), flush=True)
    if a.procs > 1:
        go = os.path.join('/tmp', f'plan_probe_go_{os.getpid()}')
        procs = []
        for r in range(a.procs):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--batch', str(a.batch), '--size', str(a.size), '--precision', a.precision,
                                           '--child', str(r), str(a.procs), go], stdout=subprocess.PIPE, text=True))
            # children record one after the other (they share this box's single GPU): wait for "ready"
            line = procs[-1].stdout.readline()
            assert line.startswith('ready'), line
        open(go, 'w').close()
        out = [json.loads(p.stdout.readline()) for p in procs]
        for p in procs:
            p.wait()
        os.unlink(go)
        ms = [o['null_ms_per_step'] for o in out]
        print(json.dumps({'contention_probe': {'procs': a.procs, 'cpus_per_proc': [o['cpus'] for o in out], 'null_backend_host_ms_per_step': ms,
                                                'alone': res['plan_null_backend_host_ms_per_step'], 'worst_over_alone': round(max(ms) / res['plan_null_backend_host_ms_per_step'], 3)}}))


def child(a):
    rank, nprocs, go = int(a.child[0]), int(a.child[1]), a.child[2]
    from oodgan.parallel import bind_rank_to_cpus
    cpus = bind_rank_to_cpus(rank, nprocs)
    parts = build(a)
    torch = parts[0]
    from oodgan import _lib
    _, run_p = make_run(*parts, use_plan=True)
    _lib.lib().oodgan_plan_set_null_launch(1)
    run_p.plan.run(20)
    print('ready', flush=True)
    while not os.path.exists(go):
        time.sleep(0.001)
    t0 = time.perf_counter()
    run_p.plan.run(400)
    ms = (time.perf_counter() - t0) * 1e3 / 400
    print(json.dumps({'rank': rank, 'cpus': len(cpus) if cpus else 0, 'null_ms_per_step': round(ms, 4)}), flush=True)


if __name__ == '__main__':
    main()
