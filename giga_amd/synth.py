"""Synthetic, seed-reproducible inputs for the GIGA hot path.

TSDF grids are U[0,1) float32 (Open3D's TSDF export range, reference perception.py:107-115);
an optional "realistic" variant zeroes ~60 % of the voxels (0 = unobserved).  Query points are
U[-0.5,0.5)^3 (optionally widened to exercise both clamps of normalize_coordinate).
"""
import numpy as np

RES = 40


def tsdf_scene(scene_idx, realistic=False):
    rng = np.random.default_rng(1234 + int(scene_idx))
    x = rng.random((RES, RES, RES), dtype=np.float32)
    if realistic:
        x = np.where(rng.random((RES, RES, RES), dtype=np.float32) < 0.6, np.float32(0), x)
    return x


def tsdf_batch(first_scene, n, realistic=False):
    return np.stack([tsdf_scene(first_scene + i, realistic) for i in range(n)], axis=0)


def tsdf_scenes(indices, realistic=False):
    """TSDF grids of an arbitrary list of scene indices (a rank's shard: scene i -> rank i mod world)."""
    return np.stack([tsdf_scene(i, realistic) for i in indices], axis=0)


def query_points_for(indices, n_points, stream=0, half_width=0.5):
    """query_points for an arbitrary list of scene indices: row k equals query_points(indices[k], 1, ...)[0]."""
    return np.concatenate([query_points(i, 1, n_points, stream, half_width) for i in indices], axis=0)


def query_points(first_scene, n_scenes, n_points, stream=0, half_width=0.5):
    """(n_scenes, n_points, 3) float32, U[-half_width, half_width)."""
    out = np.empty((n_scenes, n_points, 3), np.float32)
    for i in range(n_scenes):
        rng = np.random.default_rng([77 + stream, first_scene + i])
        out[i] = (rng.random((n_points, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(2 * half_width)
    return out


def inference_lattice(resolution=RES):
    """The exact query lattice of reference detection_implicit.py:28-31, (1, R^3, 3) float32.

    torch.linspace(-0.5, 0.5-1/R, R) evaluated the way torch does for float32 (symmetric
    two-sided formula) so the lattice is bit-identical to the reference's.
    """
    import torch

    lin = torch.linspace(-0.5, 0.5 - 1.0 / resolution, resolution)
    x, y, z = torch.meshgrid(lin, lin, lin, indexing="ij")
    return torch.stack((x, y, z), dim=-1).float().reshape(1, resolution ** 3, 3).numpy()


def train_labels(first_scene, n_scenes, n_occ):
    """Labels in train_giga.prepare_batch shapes (train_giga.py:141-151)."""
    rng = np.random.default_rng([991, first_scene])
    label = (rng.random(n_scenes) < 0.5).astype(np.float32)
    q = rng.normal(size=(n_scenes, 2, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    width = rng.uniform(0.0, 0.3, size=n_scenes).astype(np.float32)
    occ = (rng.random((n_scenes, n_occ)) < 0.3).astype(np.float32)
    return label, q, width, occ


def post_volumes(seed, R=RES):
    """Synthetic inputs for the grasp post-processing (detection_implicit.py:115-174): a TSDF with empty,
    near-surface and free-space regions, a smooth quality field with a few strong peaks, unit quaternions and
    widths in [0, 0.3]."""
    rng = np.random.default_rng([4321, seed])
    g = np.stack(np.meshgrid(*(np.linspace(-1, 1, R),) * 3, indexing="ij"), -1)
    blob = np.zeros((R, R, R))
    for _ in range(6):
        c = rng.uniform(-0.6, 0.6, 3)
        blob += np.exp(-((g - c) ** 2).sum(-1) / rng.uniform(0.02, 0.08))
    tsdf = np.clip(1.0 - blob, 0.0, 1.0).astype(np.float32)
    tsdf[rng.random((R, R, R)) < 0.15] = 0.0                       # unobserved voxels
    qual = rng.random((R, R, R))
    for _ in range(12):
        c = rng.uniform(-0.7, 0.7, 3)
        qual += 6.0 * np.exp(-((g - c) ** 2).sum(-1) / 0.01)
    qual = (1.0 / (1.0 + np.exp(-(qual - 1.5)))).astype(np.float32)
    rot = rng.standard_normal((R, R, R, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    width = rng.uniform(0.0, 0.3, (R, R, R)).astype(np.float32)
    return tsdf, qual, rot, width
