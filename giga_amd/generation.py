"""Occupancy / geometry query path: counterpart of `Generator3D.eval_points` and the dense-grid part of
`generate_from_latent` (/root/reference/src/vgn/ConvONets/conv_onet/generation.py:326-358, 144-190), SURVEY.md
section 8f-2.

The reference splits the query set into `points_batch_size` chunks, moves each chunk to the device, calls
`model.decode_occ(pi, c).logits` and copies every chunk back.  Here the planes stay cached on the device in the
layout the decoder reads (`PlaneDict.nhwc`, produced once by `encode_inputs`), the fused decoder takes a whole
chunk in ONE launch (there is no activation tensor whose size would call for chunking: `points_batch_size` only bounds
the device copy of a HOST-resident query set, one chunk in flight at a time), and a regular grid of
queries takes the lattice path (each plane sampled once per lattice coordinate pair).  Mesh extraction
(`extract_mesh`, marching cubes via libmcubes) is host code outside the hot path and not part of this package."""
import torch

from . import _capi
from .convonet import register_lattice


class Generator3D:
    """generation.py:22-75 (the arguments that matter for the voxel-input GIGA models)."""

    def __init__(self, model, points_batch_size=100000, threshold=0.5, device=None, resolution0=16, **mesh_options):
        """`threshold`, `upsampling_steps`, `padding`, ... of generation.py:36-58 steer the host-side mesh extraction
        (marching cubes / MISE), which is outside this package: they are accepted and kept in `mesh_options` untouched."""
        self.model = model
        self.points_batch_size = int(points_batch_size)
        self.threshold = threshold
        self.device = device if device is not None else torch.device("cuda")
        self.resolution0 = resolution0
        self.mesh_options = dict(mesh_options)
        self._grids = {}

    def encode(self, inputs):
        """generation.py:99-101: c = model.encode_inputs(inputs), kept on the device for every later query."""
        with torch.no_grad():
            return self.model.encode_inputs(inputs.to(self.device))

    def eval_points(self, p, c=None, **kwargs):
        """generation.py:326-358 (`else` branch: voxel / point-cloud inputs).  p: (N,3) or (B,N,3) points in the
        unit cube, on any device; returns the occupancy logits on the device, (N,) or (B,N).  One launch; the
        reference's chunk loop and per-chunk D->H copy disappear (use `.cpu()` for its return type)."""
        squeeze = p.dim() == 2
        pts = p.unsqueeze(0) if squeeze else p
        with torch.no_grad():
            if pts.is_cuda or pts.shape[1] <= self.points_batch_size:
                logits = self.model.decode_occ(pts.to(self.device, torch.float32), c, **kwargs).logits
            else:                                             # host-resident queries: bounded device footprint (generation.py:337)
                parts = [self.model.decode_occ(ch.to(self.device, torch.float32), c, **kwargs).logits
                         for ch in torch.split(pts, self.points_batch_size, dim=1)]
                logits = torch.cat(parts, dim=1)
        return logits.squeeze(0) if squeeze else logits

    def grid_points(self, resolution, lo=-0.5, hi=0.5):
        """The (1, R^3, 3) regular grid `box_size * make_3d_grid((-0.5,)*3, (0.5,)*3, (R,)*3)` of generation.py:160-165
        (box_size folded into lo/hi), registered as a lattice so that decoding it takes the lattice path."""
        key = (int(resolution), float(lo), float(hi))
        g = self._grids.get(key)
        if g is None:
            lin = torch.linspace(lo, hi, resolution)
            x, y, z = torch.meshgrid(lin, lin, lin, indexing="ij")
            pts = torch.stack((x, y, z), dim=-1).float().reshape(1, resolution ** 3, 3).to(self.device)
            g = self._grids[key] = register_lattice(pts, lin)
        return g

    def occupancy_grid(self, c, resolution=None, lo=-0.5, hi=0.5):
        """generation.py:157-166 (upsampling_steps == 0): logits on the dense R^3 grid -> (B,R,R,R) device tensor."""
        if resolution is None:
            resolution = self.resolution0
        if resolution > 64:
            raise _capi.GigaHipError("the lattice decoder supports up to 64 points per axis; use eval_points")
        with torch.no_grad():
            logits = self.model.decode_occ(self.grid_points(resolution, lo, hi), c).logits
        return logits.view(-1, resolution, resolution, resolution)
