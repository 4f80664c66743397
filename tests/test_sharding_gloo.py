"""world_size=2 `gloo` test of the N>1 path on CPU: scene sharding + output gather + counters.
The compute callable is injected; here it is the CPU oracle (no GPU in this container), on the GPU
box it is a giga_amd network.  Verifies that sharded+gathered == single-process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from giga_amd import sharding, synth, weights
from oracle import giga_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_scenes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = weights.make_state_dict(7)
    x = torch.from_numpy(synth.tsdf_batch(0, n_scenes))
    p = torch.from_numpy(synth.query_points(0, n_scenes, 16, stream=1))
    pt = torch.from_numpy(synth.query_points(0, n_scenes, 24, stream=2))

    def fwd(xl, pl, ptl):
        with torch.no_grad():
            return O.model_forward(sd, xl, pl, p_tsdf=ptl)

    out = sharding.run_sharded(fwd, x, p, pt)
    counters = sharding.gather_counters([len(sharding.scene_shard(n_scenes, rank, world)), float(rank)])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), qual=out[0].numpy(), rot=out[1].numpy(),
             width=out[2].numpy(), tsdf=out[3].numpy(), counters=counters.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from giga_amd.training import allreduce_mean_
    flat = torch.arange(581863, dtype=torch.float32) * (rank + 1)          # the flat gradient bucket
    allreduce_mean_(flat)
    np.save(os.path.join(out_dir, f"g{rank}.npy"), flat[:1000].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_gradient_allreduce_gloo(tmp_path):
    """Data-parallel training's only collective: mean of the flat gradient bucket over 2 ranks."""
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    expect = np.arange(1000, dtype=np.float32) * 1.5
    for r in range(2):
        np.testing.assert_allclose(np.load(os.path.join(tmp_path, f"g{r}.npy")), expect, rtol=1e-6)


def test_shard_maps():
    assert sharding.scene_shard(5, 0, 2) == [0, 2, 4] and sharding.scene_shard(5, 1, 2) == [1, 3]
    assert sharding.shard_sizes(256, 8) == [32] * 8
    assert sorted(sum((sharding.scene_shard(11, r, 4) for r in range(4)), [])) == list(range(11))
    assert sharding.scene_shard(1, 3, 8) == []


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    n_scenes, world = 3, 2                     # ragged: rank 0 owns 2 scenes, rank 1 owns 1
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_scenes, str(tmp_path)), nprocs=world, join=True)
    sd = weights.make_state_dict(7)
    x = torch.from_numpy(synth.tsdf_batch(0, n_scenes))
    p = torch.from_numpy(synth.query_points(0, n_scenes, 16, stream=1))
    pt = torch.from_numpy(synth.query_points(0, n_scenes, 24, stream=2))
    with torch.no_grad():
        ref = O.model_forward(sd, x, p, p_tsdf=pt)
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for name, t in zip(("qual", "rot", "width", "tsdf"), ref):
            assert got[name].shape == tuple(t.shape)
            assert np.abs(got[name] - t.numpy()).max() < 1e-5, (r, name)
        assert got["counters"].tolist() == [[2.0, 0.0], [1.0, 1.0]]
