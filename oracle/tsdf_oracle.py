"""TEST INFRASTRUCTURE (CPU oracle; only tests/, smoke() and bench.py's cpu_baseline leg may import oracle/).

Restatement of `TSDFVolume.get_grid` (reference src/vgn/perception.py:107-115): the dense (1, R, R, R) float32 export of
Open3D's sparse voxel list.  Open3D is absent from this image, so the voxel list itself (`extract_voxel_grid().get_voxels()`)
cannot be produced here: PARITY UNPINNED for the Open3D call; the loop that follows it is plain numpy indexing and is
restated verbatim in semantics (zeros elsewhere, later voxels overwrite earlier ones)."""
import numpy as np


def get_grid(voxel_index, voxel_value, resolution=40):
    """perception.py:109-115 with the voxel list given as arrays: grid_index (n,3) and color[0] (n,)."""
    shape = (1, resolution, resolution, resolution)
    tsdf_grid = np.zeros(shape, dtype=np.float32)
    for (i, j, k), c in zip(np.asarray(voxel_index).reshape(-1, 3), np.asarray(voxel_value).reshape(-1)):
        tsdf_grid[0, i, j, k] = c
    return tsdf_grid


def synthetic_voxels(seed, resolution=40, fill=0.35, duplicates=0):
    """A seeded sparse voxel list shaped like Open3D's (unique indices, values in (0, 1]); `duplicates` appends repeated
    indices with new values to exercise the last-one-wins rule of the loop."""
    rng = np.random.default_rng([2024, seed])
    R = resolution
    cells = rng.permutation(R * R * R)[: int(fill * R * R * R)]
    idx = np.stack((cells // (R * R), (cells // R) % R, cells % R), -1).astype(np.int32)
    val = (1.0 - rng.random(idx.shape[0])).astype(np.float32)
    if duplicates:
        pick = rng.integers(0, idx.shape[0], duplicates)
        idx = np.concatenate([idx, idx[pick]])
        val = np.concatenate([val, (1.0 - rng.random(duplicates)).astype(np.float32)])
    return idx, val
