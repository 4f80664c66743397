"""Synthetic, seed-reproducible inputs for the GIGA hot path.

TSDF grids are U[0,1) float32 (Open3D's TSDF export range, reference perception.py:107-115);
an optional "realistic" variant zeroes ~60 % of the voxels (0 = unobserved).  Query points are
U[-0.5,0.5)^3 (optionally widened to exercise both clamps of normalize_coordinate).
"""
import json
import os

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


# ---- a synthetic training set in the reference's on-disk layout ---------------------------------------------------
# scenes/<id>.npz ("grid"), grasps.csv, setup.json, occ/<id>/*.npz ("points", "occ"): the files vgn.io.write_voxel_grid
# (io.py:88-90), write_grasp (:56-69), write_setup (:11-17) and the occupancy generator produce.  Read back by
# giga_amd.dataset.GraspOccDataset (and by the reference's DatasetVoxelOccFile in the parity tests).
def write_training_set(root, raw_root, n_scenes=6, grasps_per_scene=5, occ_files=(1, 3), n_occ_points=300, seed=0, size=0.3):
    rng = np.random.default_rng([606, seed])
    os.makedirs(os.path.join(root, "scenes"), exist_ok=True)
    os.makedirs(raw_root, exist_ok=True)
    rows = []
    for s in range(n_scenes):
        sid = f"scene{seed:02d}_{s:04d}"
        np.savez_compressed(os.path.join(root, "scenes", sid + ".npz"), grid=tsdf_scene(1000 * seed + s, realistic=True)[None])
        d = os.path.join(raw_root, "occ", sid)
        os.makedirs(d, exist_ok=True)
        for f in range(int(rng.integers(occ_files[0], occ_files[1] + 1))):
            pts = (rng.random((n_occ_points, 3)) * size).astype(np.float32)
            np.savez(os.path.join(d, f"{f:04d}.npz"), points=pts, occ=rng.random(n_occ_points) < 0.3)
        for _ in range(grasps_per_scene):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            rows.append((sid, *q, *(rng.random(3) * size), rng.uniform(0.02, 0.08), int(rng.random() < 0.5)))
    order = rng.permutation(len(rows))
    with open(os.path.join(raw_root, "grasps.csv"), "w") as f:
        f.write("scene_id,qx,qy,qz,qw,x,y,z,width,label\n")
        for i in order:
            r = rows[i]
            f.write(",".join([r[0]] + [repr(float(v)) for v in r[1:9]] + [str(r[9])]) + "\n")
    with open(os.path.join(raw_root, "setup.json"), "w") as f:
        json.dump({"size": size, "intrinsic": {"width": 640, "height": 480, "K": [540.0, 0.0, 320.0, 0.0, 540.0, 240.0, 0.0, 0.0, 1.0]},
                   "max_opening_width": 0.08, "finger_depth": 0.05}, f)
    return len(rows)
