"""SURVEY 8f-3, producer half: device-side dense export of a sparse TSDF voxel list against the oracle's restatement of
the reference loop (perception.py:107-115).  Bit-exact (it is a scatter of float32 values)."""
import numpy as np
import pytest
import torch

from oracle import tsdf_oracle as TO


def test_oracle_loop_semantics():
    idx = np.array([[0, 0, 0], [1, 2, 3], [1, 2, 3], [39, 39, 39]], np.int32)
    val = np.array([0.5, 0.25, 0.75, 1.0], np.float32)
    g = TO.get_grid(idx, val)
    assert g.shape == (1, 40, 40, 40) and g.dtype == np.float32
    assert g[0, 0, 0, 0] == 0.5 and g[0, 1, 2, 3] == 0.75 and g[0, 39, 39, 39] == 1.0      # the last duplicate wins
    assert np.count_nonzero(g) == 3
    idx2, val2 = TO.synthetic_voxels(3)
    assert len(np.unique(idx2 @ np.array([1600, 40, 1]))) == len(idx2)                        # unique like Open3D's list


def test_host_index_validation_without_gpu():
    from giga_amd import _capi, perception
    with pytest.raises(_capi.GigaHipError):
        perception.dense_grid(np.zeros((1, 3), np.int32), np.ones(1, np.float32), device="cpu")
    lib = _capi.lib()
    assert lib.giga_tsdf_scatter_workspace_bytes(2, 40) == 2 * 64000 * 4
    assert lib.giga_tsdf_scatter(None, None, None, 0, 40, 0, None, None, 0, None) == 0       # empty batch
    assert lib.giga_tsdf_scatter(None, None, None, 1, 40, 5, None, None, 0, None) == -1


@pytest.mark.gpu
def test_dense_export_matches_reference_loop():
    from giga_amd import perception
    dev = torch.device("cuda:0")
    scenes = [TO.synthetic_voxels(s, duplicates=d) for s, d in ((0, 0), (1, 500), (2, 0))]
    scenes.append((np.zeros((0, 3), np.int32), np.zeros(0, np.float32)))                      # an empty (unobserved) scene
    got = perception.dense_grids(scenes, 40, dev).cpu().numpy()
    assert got.shape == (4, 40, 40, 40)
    for b, (idx, val) in enumerate(scenes):
        assert np.array_equal(got[b], TO.get_grid(idx, val)[0]), b
    one = perception.dense_grid(*scenes[1], device=dev)
    assert one.shape == (1, 40, 40, 40) and np.array_equal(one.cpu().numpy(), TO.get_grid(*scenes[1]))
    with pytest.raises(IndexError):
        perception.dense_grid(np.array([[0, 40, 0]], np.int32), np.ones(1, np.float32), device=dev)

    class V:                                                   # Open3D-style voxel objects
        def __init__(self, gi, c):
            self.grid_index, self.color = gi, (c, c, c)
    vox = [V(tuple(i), float(c)) for i, c in zip(*scenes[0])][:2000]
    idx, val = perception.voxel_arrays(vox)
    assert np.array_equal(perception.dense_grid(idx, val, device=dev).cpu().numpy(), TO.get_grid(idx, val))
    # the export feeds the encoder directly: same planes as from a host-built grid
    from giga_amd import networks, weights
    net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
    with torch.no_grad():
        a = net.encode_inputs(perception.dense_grids(scenes[:2], 40, dev))
        b = net.encode_inputs(torch.from_numpy(np.stack([TO.get_grid(*s)[0] for s in scenes[:2]])).to(dev))
    assert all(torch.equal(a[k], b[k]) for k in a)
