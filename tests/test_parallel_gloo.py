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
