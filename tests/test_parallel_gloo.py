"""world_size-2 gloo test of the batch-sharding path (runs on CPU): the shard slices tile the
global batch, the per-rank results gathered with all_gather equal the unsharded result, ragged and
empty shards included."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_invert(target, w0, noises):
    # stands in for the GPU inversion: any per-image function of this rank's slice.  Like the HIP entry points it
    # refuses an empty batch — invert_sharded must not call it for an empty shard
    if w0.shape[0] == 0:
        raise RuntimeError('liboodgan_hip: B must be > 0')
    return w0 * 2.0 + target.mean(dim=(1, 2, 3)).view(-1, 1, 1) + noises[0].sum(dim=(1, 2, 3)).view(-1, 1, 1)


def _worker(rank, world, port, gB, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'ood-gan-inversion_amd'))
    import torch.distributed as dist
    from oodgan import parallel, synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    target = synth.make_images(8, gB, seed=1)
    w0 = synth.make_latents(16, gB, seed=3)
    noises = [synth.normal('n', (gB, 1, 4, 4), 2)]
    full = _fake_invert(target, w0, noises) if gB else w0
    got = parallel.invert_sharded(_fake_invert, dict(target=target, w0=w0, noises=noises), gB, rank, world)
    ok = got.shape == full.shape and torch.equal(got, full)
    got2 = parallel.gather_latents(full[parallel.shard_slice(gB, rank, world)])   # size discovery path
    ok = ok and torch.equal(got2, full)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _worker8(rank, world, port, gB, fail_rank, q):
    """BASELINE configs[3] geometry: global batch 64 over 8 ranks; ``fail_rank`` raises inside its inversion."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'ood-gan-inversion_amd'))
    import torch.distributed as dist
    from oodgan import parallel, synth
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    target = synth.make_images(8, gB, seed=1)
    w0 = synth.make_latents(1024, gB, seed=3)                       # (64, 18, 512): the real latent shape
    noises = [synth.normal('n', (gB, 1, 4, 4), 2)]
    full = _fake_invert(target, w0, noises)

    def inv(target, w0, noises):
        if rank == fail_rank:
            raise RuntimeError('liboodgan_hip: injected failure')
        return _fake_invert(target, w0, noises)

    sl = parallel.shard_slice(gB, rank, world)
    try:
        got = parallel.invert_sharded(inv, dict(target=target, w0=w0, noises=noises), gB, rank, world)
        res = ('ok', bool(got.shape == full.shape and torch.equal(got, full)))
    except parallel.ShardFailed as e:
        fs = parallel.shard_slice(gB, fail_rank, world)
        good = torch.ones(gB, dtype=torch.bool)
        good[fs] = False
        res = ('peer', e.ranks == [fail_rank] and bool(torch.isnan(e.latents[fs]).all()) and torch.equal(e.latents[good], full[good]))
    except RuntimeError as e:
        res = ('own', 'injected failure' in str(e))
    q.put((rank, sl.stop - sl.start) + res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('fail_rank', [-1, 5])
def test_world8_global_batch_64(fail_rank):
    """8 ranks x 8 images, one all_gather; with a failing rank EVERY rank raises (its own error / ShardFailed) after the
    collective and nobody returns NaN rows silently."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, 64, fail_rank, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [8] * 8
    for rank, _, kind, ok in res:
        assert ok, res
        assert kind == ('ok' if fail_rank < 0 else 'own' if rank == fail_rank else 'peer'), res


def test_cpu_slices_for_ranks():
    from oodgan.parallel import _parse_cpulist, bind_rank_to_cpus, cpu_slice_for_rank
    cpus = list(range(128))
    parts = [cpu_slice_for_rank(cpus, r, 8) for r in range(8)]
    assert all(len(p) == 16 for p in parts) and sorted(sum(parts, [])) == cpus            # disjoint, covering
    assert cpu_slice_for_rank(cpus, 3, 8, numa_cpus=range(64, 128)) == list(range(64, 128))
    assert cpu_slice_for_rank([0, 1], 5, 8) == [0, 1]                                       # more ranks than CPUs: never empty
    assert cpu_slice_for_rank(cpus, 0, 8, numa_cpus=[500]) == list(range(16))               # NUMA list outside the allowed set
    assert _parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert bind_rank_to_cpus(0, 1) is None                                                  # single process: not bound
    before = sorted(os.sched_getaffinity(0))
    import torch
    nt = torch.get_num_threads()
    try:
        got = bind_rank_to_cpus(1, 2)
        assert got and set(got) <= set(before) and sorted(os.sched_getaffinity(0)) == sorted(got)
        assert torch.get_num_threads() == len(got)          # the intra-op pool follows the slice (8 ranks x all cores otherwise)
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(nt)


@pytest.mark.parametrize('gB', [4, 5, 1])
def test_sharded_inversion_matches_unsharded(gB):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gB, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_shard_slices_tile_the_batch():
    from oodgan.parallel import shard_slice, shard_sizes
    for gB in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            idx = []
            for r in range(world):
                s = shard_slice(gB, r, world)
                idx += list(range(s.start, s.stop))
            assert idx == list(range(gB))
            assert sum(shard_sizes(gB, world)) == gB
            assert max(shard_sizes(gB, world)) - min(shard_sizes(gB, world)) <= 1
