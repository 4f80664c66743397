"""Scene-level data parallelism.

Scenes are independent in every function of the hot path (the batch dimension is carried untouched
from the TSDF to the head outputs), so multi-GPU execution is one process per GPU, scene i -> rank
i mod world, replicated weights (2.3 MB), and NO data-path collective.  The only communication is an
optional all_gather (RCCL over xGMI with backend 'nccl'; 'gloo' in the CPU tests) that reassembles
per-scene outputs / collects throughput counters on every rank (BASELINE.json configs[2]).
The reference itself is single-device (train_giga.py:20-21, detection_implicit.py:19)."""
import torch
import torch.distributed as dist


def scene_shard(n_scenes, rank, world):
    """Indices of the scenes owned by `rank`:  i -> rank i mod world."""
    return list(range(rank, n_scenes, world))


def shard_sizes(n_scenes, world):
    return [len(range(r, n_scenes, world)) for r in range(world)]


def all_gather_scenes(local, n_scenes, rank=None, world=None, group=None):
    """Reassemble per-scene tensors computed under `scene_shard` into global scene order.

    local: tensor (n_local, ...) or tuple of such.  One all_gather per tensor; shards are padded to
    the largest shard so the collective has equal shapes on every rank."""
    if isinstance(local, (tuple, list)):
        return tuple(all_gather_scenes(t, n_scenes, rank, world, group) for t in local)
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    sizes = shard_sizes(n_scenes, world)
    cap = max(sizes)
    pad = local.new_zeros((cap,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = local.new_empty((n_scenes,) + tuple(local.shape[1:]))
    for r in range(world):
        idx = scene_shard(n_scenes, r, world)
        out[idx] = parts[r][: sizes[r]]
    return out


def run_sharded(forward_fn, tsdf, p, p_tsdf=None, rank=None, world=None, group=None, gather=True):
    """Evaluate `forward_fn(tsdf_local, p_local, p_tsdf_local)` on this rank's scenes and (optionally)
    gather every rank's outputs into global order.  `forward_fn` is e.g. a giga_amd network on this
    rank's GPU; tensors are indexed on whatever device they live on."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    n = tsdf.shape[0]
    idx = scene_shard(n, rank, world)
    args = [tsdf[idx].contiguous(), p[idx].contiguous()]
    if p_tsdf is not None:
        args.append(p_tsdf[idx].contiguous())
    local = forward_fn(*args)
    if not gather:
        return local
    return all_gather_scenes(tuple(local), n, rank, world, group)


def gather_counters(values, group=None):
    """all_gather of a small float64 vector of per-rank counters (scenes done, seconds, ...)."""
    t = torch.as_tensor(values, dtype=torch.float64)
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        t = t.cuda()
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    return torch.stack(parts).cpu()
