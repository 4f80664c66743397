"""CPU oracle for the grasp post-processing that follows the network in the reference's planner
(src/vgn/detection_implicit.py:87-174: process, bound, select).  TEST INFRASTRUCTURE ONLY (see
oracle/giga_oracle.py).  Uses the same scipy.ndimage primitives the reference calls; pinned against the
reference functions themselves by oracle/make_post_goldens.py -> tests/golden/g6_postprocess.npz."""
import numpy as np
from scipy import ndimage

LOW_TH = 0.5          # detection_implicit.py:15


def process(tsdf_vol, qual_vol, width_vol, gaussian_filter_sigma=1.0, min_width=0.033, max_width=0.233, out_th=0.5):
    """detection_implicit.py:115-143 (rot_vol is passed through untouched by the reference)."""
    tsdf_vol = tsdf_vol.squeeze()
    qual_vol = ndimage.gaussian_filter(qual_vol, sigma=gaussian_filter_sigma, mode="nearest")
    outside_voxels = tsdf_vol > out_th
    inside_voxels = np.logical_and(1e-3 < tsdf_vol, tsdf_vol < out_th)
    valid_voxels = ndimage.binary_dilation(outside_voxels, iterations=2, mask=np.logical_not(inside_voxels))
    qual_vol[valid_voxels == False] = 0.0  # noqa: E712
    qual_vol[np.logical_or(width_vol < min_width, width_vol > max_width)] = 0.0
    return qual_vol


def bound_limits(voxel_size, limit=(0.02, 0.02, 0.055)):
    """detection_implicit.py:87-90."""
    return tuple(int(l / voxel_size) for l in limit)


def bound(qual_vol, voxel_size, limit=(0.02, 0.02, 0.055)):
    """detection_implicit.py:87-97."""
    x_lim, y_lim, z_lim = bound_limits(voxel_size, limit)
    qual_vol[:x_lim] = 0.0
    qual_vol[-x_lim:] = 0.0
    qual_vol[:, :y_lim] = 0.0
    qual_vol[:, -y_lim:] = 0.0
    qual_vol[:, :, :z_lim] = 0.0
    return qual_vol


def select_indices(qual_vol, threshold=0.90, max_filter_size=4, force_detection=False):
    """detection_implicit.py:146-174 up to the Grasp objects: (indices (K,3) sorted by descending score, scores)."""
    qual_vol = qual_vol.copy()
    best_only = False
    qual_vol[qual_vol < LOW_TH] = 0.0
    if force_detection and (qual_vol >= threshold).sum() == 0:
        best_only = True
    else:
        qual_vol[qual_vol < threshold] = 0.0
    max_vol = ndimage.maximum_filter(qual_vol, size=max_filter_size)
    qual_vol = np.where(qual_vol == max_vol, qual_vol, 0.0)
    idx = np.argwhere(np.where(qual_vol, 1.0, 0.0))
    scores = np.array([qual_vol[i, j, k] for i, j, k in idx], dtype=np.float32)
    order = list(reversed(np.argsort(scores)))
    idx, scores = idx[order], scores[order]
    if best_only and len(idx) > 0:
        idx, scores = idx[:1], scores[:1]
    return idx.reshape(-1, 3), scores
