"""Input side of the hot path: dense export of a TSDF volume on the device.

Counterpart of `vgn.perception.TSDFVolume.get_grid` (reference src/vgn/perception.py:107-115), which fills a
(1, R, R, R) float32 grid from Open3D's sparse voxel list with a Python loop over the voxels (the reference notes
"very slow (~35 ms / 50 ms of the whole pipeline)").  Here the sparse list is uploaded once (16 B per voxel) and a HIP
kernel scatters it into the dense grid the encoder consumes, for any number of scenes per call (`giga_tsdf_scatter`)."""
import numpy as np
import torch

from . import _capi


def voxel_arrays(voxels):
    """(index (n,3) int32, value (n,) float32) from an iterable of Open3D-style voxels (`.grid_index`, `.color`): the two
    attributes the reference loop reads (perception.py:112-114)."""
    idx = np.asarray([v.grid_index for v in voxels], dtype=np.int32).reshape(-1, 3)
    val = np.asarray([v.color[0] for v in voxels], dtype=np.float32).reshape(-1)
    return idx, val


def dense_grids(scenes, resolution=40, device=None):
    """scenes: list of (index (n_b,3) int, value (n_b,) float) per scene (numpy arrays or tensors).
    Returns a (B, R, R, R) float32 device tensor: 0 where unobserved, else the value of the last listed voxel of the cell."""
    device = torch.device(device if device is not None else "cuda")
    if device.type != "cuda":
        raise _capi.GigaHipError("dense_grids runs on a HIP device; use the reference's get_grid on the CPU")
    R, B = int(resolution), len(scenes)
    idx_l, val_l, offs = [], [], [0]
    for idx, val in scenes:
        idx = torch.as_tensor(idx).reshape(-1, 3)
        val = torch.as_tensor(val).reshape(-1)
        if idx.shape[0] != val.shape[0]:
            raise ValueError("one value per voxel index")
        if not idx.is_cuda and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= R):
            raise IndexError(f"voxel grid_index outside [0, {R})")       # numpy raises (or wraps negatives) in the reference loop
        idx_l.append(idx.to(torch.int32)); val_l.append(val.to(torch.float32))
        offs.append(offs[-1] + idx.shape[0])
    grid = torch.empty((B, R, R, R), dtype=torch.float32, device=device)
    if B == 0:
        return grid
    n = offs[-1]
    index = torch.cat(idx_l).to(device).contiguous() if n else None
    value = torch.cat(val_l).to(device).contiguous() if n else None
    offsets = torch.tensor(offs, dtype=torch.int32).to(device)
    L = _capi.lib()
    ws = torch.empty(max(L.giga_tsdf_scatter_workspace_bytes(B, R), 16), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _capi.check(L.giga_tsdf_scatter(_capi.ptr(index), _capi.ptr(value), _capi.ptr(offsets), B, R, n, _capi.ptr(grid),
                                        _capi.ptr(ws), ws.numel(), _capi.stream_ptr(device)), "giga_tsdf_scatter")
    return grid


def dense_grid(voxel_index, voxel_value, resolution=40, device=None):
    """One scene, in the reference's return shape (1, R, R, R) (perception.py:109)."""
    return dense_grids([(voxel_index, voxel_value)], resolution, device)
