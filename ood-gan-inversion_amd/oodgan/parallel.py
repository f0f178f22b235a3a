"""Batch sharding of the inversion loop over the GPUs of one node.

Every image's inversion is independent (per-image loss, per-sample InstanceNorm, frozen replicated
weights), so the global batch is cut into contiguous per-rank slices and NO collective runs during
the W+ steps.  The single exchange is one all_gather of the finished latents (B_local,18,512) fp32
= 36 KB per image — RCCL over xGMI when the process group backend is 'nccl', gloo in the CPU tests."""
import torch


def shard_slice(global_batch, rank, world_size):
    """Contiguous slice of the global batch owned by ``rank`` (ragged batches: the first
    ``global_batch % world_size`` ranks get one extra image; empty slices are allowed)."""
    q, r = divmod(global_batch, world_size)
    start = rank * q + min(rank, r)
    return slice(start, start + q + (1 if rank < r else 0))


def shard_sizes(global_batch, world_size):
    return [shard_slice(global_batch, r, world_size).stop - shard_slice(global_batch, r, world_size).start
            for r in range(world_size)]


def gather_latents(local, global_batch=None, group=None):
    """all_gather of per-rank latents into the global (B_global, L, S) tensor on every rank.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if global_batch is None:
        n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = shard_sizes(global_batch, world)
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def gather_latents_status(local, ok, global_batch, group=None):
    """ONE all_gather carrying the latents AND a per-rank status: every rank's buffer gets one extra row whose first
    element is 0.0 (shard inverted) or 1.0 (``invert_fn`` raised on that rank).  Returns (latents, [failed ranks])."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local, ([] if ok else [0])
    world = dist.get_world_size(group)
    sizes = shard_sizes(global_batch, world)
    mx = max(sizes)
    buf = local.new_zeros((mx + 1,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    buf[mx].view(-1)[0] = 0.0 if ok else 1.0
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf.contiguous(), group=group)
    failed = [r for r, b in enumerate(bufs) if float(b[mx].view(-1)[0].item()) != 0.0]
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0), failed


class ShardFailed(RuntimeError):
    """Raised on EVERY rank when the inversion of at least one shard raised; ``.ranks`` lists them, ``.latents`` holds the
    gathered tensor (NaN rows for the failed shards)."""

    def __init__(self, ranks, latents):
        super().__init__(f'invert_sharded: the inversion failed on rank(s) {ranks}; their rows of the gathered latents are NaN')
        self.ranks, self.latents = ranks, latents


def invert_sharded(invert_fn, inputs, global_batch, rank, world_size, group=None, empty_like=None):
    """Run ``invert_fn(**local_inputs) -> latents`` on this rank's slice of every tensor in
    ``inputs`` (tensors or lists of tensors with the batch on dim 0) and gather the latents.

    A rank whose slice is empty (world_size > global_batch) does NOT call ``invert_fn`` (the HIP entry points require
    B > 0): it joins the collective with an empty (0, L, S) tensor shaped like ``inputs[empty_like]`` (default: the
    'w0' / 'lats' / 'enc_lats' entry).  A rank whose ``invert_fn`` raises still joins the collective — with NaN latents and
    its status row set — so that its peers are not left blocked in all_gather; afterwards EVERY rank raises: the failing
    rank its own exception, the others ``ShardFailed`` (no rank returns NaN rows silently).  If even the NaN placeholder
    cannot be built (the device itself is gone) the original exception propagates at once and the process exits non-zero;
    the peers then fail in the collective."""
    sl = shard_slice(global_batch, rank, world_size)

    def cut(v):
        if isinstance(v, (list, tuple)):
            return [cut(t) for t in v]
        return v[sl] if isinstance(v, torch.Tensor) and v.shape[0] == global_batch else v

    local = {k: cut(v) for k, v in inputs.items()}
    key = empty_like or next((k for k in ('w0', 'lats', 'enc_lats') if isinstance(inputs.get(k), torch.Tensor)), None)
    err = None
    if sl.stop == sl.start:
        if key is None:
            raise ValueError('invert_sharded: an empty shard needs a latent-shaped input (w0 / lats / enc_lats or empty_like=)')
        lat = local[key][:0]
    else:
        try:
            lat = invert_fn(**local)
        except Exception as e:          # noqa: BLE001 — re-raised below, after the collective
            if key is None:
                raise
            err = e
            try:
                lat = torch.full_like(local[key], float('nan'))
            except Exception:           # noqa: BLE001 — the device is unusable: nothing to join the collective with
                raise e
    out, failed = gather_latents_status(lat, err is None, global_batch, group)
    if err is not None:
        raise err
    if failed:
        raise ShardFailed(failed, out)
    return out


# ------------------------------------------------------------------------------------- CPU placement of the ranks
def cpu_slice_for_rank(cpus, local_rank, local_world, numa_cpus=None):
    """The host CPUs one rank's process (launch thread + torch's helper threads) is bound to.  ``cpus``: the CPUs this
    process may run on; ``numa_cpus``: the CPUs of the GPU's NUMA node, or None when unknown.  Ranks whose GPUs share a
    node split that node's CPUs evenly by their order on it; without NUMA information the allowed set is cut into
    ``local_world`` contiguous slices.  Never returns an empty set."""
    cpus = sorted(cpus)
    if numa_cpus:
        mine = [c for c in cpus if c in set(numa_cpus)]
        if mine:
            return mine
    q = max(1, len(cpus) // max(1, local_world))
    part = cpus[local_rank * q:(local_rank + 1) * q]
    return part or cpus


def _gpu_numa_nodes():
    """NUMA node of every GPU in KFD topology order (the order HIP enumerates them in), read from sysfs — no HIP call,
    so this is safe before the process has touched the GPU."""
    import glob
    import os
    nodes = []
    for d in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/*'), key=lambda p: int(os.path.basename(p))):
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(d, 'properties')) if len(line.split()) >= 2)
            if int(props.get('simd_count', 0)) == 0:
                continue                                   # a CPU node
            loc, dom = int(props.get('location_id', 0)), int(props.get('domain', 0))
            bdf = f'{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}'
            with open(f'/sys/bus/pci/devices/{bdf}/numa_node') as f:
                nodes.append(int(f.read()))
        except Exception:                                  # noqa: BLE001
            nodes.append(-1)
    return nodes


def bind_rank_to_cpus(local_rank=None, local_world=None):
    """Pin this process to host CPUs near its GPU, chosen from LOCAL_RANK — call BEFORE ``torch.cuda.set_device`` (it is a
    plain ``sched_setaffinity``: no re-exec, nothing touches the GPU).  Eight ranks launching ~170 kernels per W+ step
    each otherwise migrate across sockets; torch's intra-op thread pool is then sized to the slice (``torch.set_num_threads``).
    Returns the CPU list (or None where affinity is not supported)."""
    import os
    if not hasattr(os, 'sched_setaffinity'):
        return None
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) if local_rank is None else local_rank
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', 1))) if local_world is None else local_world
    if local_world <= 1:
        return None                                            # a single process keeps the whole host
    cpus = sorted(os.sched_getaffinity(0))
    numa_cpus = None
    try:
        vis = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')
        nodes = _gpu_numa_nodes()
        if vis:
            order = [int(v) for v in vis.split(',') if v.strip().isdigit()]
            nodes = [nodes[i] for i in order if i < len(nodes)]
        node = nodes[local_rank] if local_rank < len(nodes) else -1
        if node >= 0:
            with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
                numa_cpus = _parse_cpulist(f.read())
            peers = [r for r in range(min(local_world, len(nodes))) if nodes[r] == node]
            mine = [c for c in cpus if c in set(numa_cpus)]
            if mine and len(peers) > 1:
                numa_cpus = cpu_slice_for_rank(mine, peers.index(local_rank), len(peers))
    except Exception:                                      # noqa: BLE001 — placement is best effort
        numa_cpus = None
    chosen = cpu_slice_for_rank(cpus, local_rank, local_world, numa_cpus)
    try:
        os.sched_setaffinity(0, chosen)
    except OSError:
        return None
    # torch sized its intra-op pool from the whole host when it was imported: eight ranks x 128 default threads would
    # oversubscribe the cores during the synthetic-input generation and the host-side bookkeeping — size it to the slice
    try:
        import torch
        torch.set_num_threads(max(1, len(chosen)))
    except Exception:                                      # noqa: BLE001
        pass
    return chosen


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out += list(range(int(a), int(b or a) + 1))
    return out
