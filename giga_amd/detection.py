"""Inference-side caller of the hot path: counterpart of the network part of the reference's
`vgn.detection_implicit` (/root/reference/src/vgn/detection_implicit.py:17-31, 99-113).

The 40^3 query lattice and `predict()` keep the reference's shapes and dtypes.  The scipy
post-processing (`process` / `bound` / `select`, detection_implicit.py:87-174) stays on the host in
the reference and is listed as "next" in SURVEY.md section 8f; it consumes exactly what `predict`
returns, so the reference functions can be used unchanged on these outputs."""
import numpy as np
import torch

from . import synth


def query_lattice(resolution=40, device=None):
    """detection_implicit.py:28-31 -> (1, R^3, 3) float32 (bit-identical to the reference's self.pos).

    The returned tensor is registered as "the inference lattice": passing it (the same object) to the
    network selects the lattice fast path of the decoder, which samples each plane once per lattice
    coordinate pair instead of once per point, and it may be shared by a whole batch of scenes."""
    from .convonet import register_lattice
    pos = torch.from_numpy(synth.inference_lattice(resolution))
    if device is not None:
        pos = pos.to(device)
    lin = pos[0, :: resolution * resolution, 0].clone()      # the R coordinates (x varies slowest)
    return register_lattice(pos, lin)


def predict(tsdf_vol, pos, net, device):
    """detection_implicit.py:99-113: tsdf_vol (1,40,40,40) numpy -> qual (64000,), rot (64000,4), width (64000,)."""
    assert tsdf_vol.shape == (1, 40, 40, 40)
    tsdf_vol = torch.from_numpy(np.ascontiguousarray(tsdf_vol, dtype=np.float32)).to(device)
    with torch.no_grad():
        qual_vol, rot_vol, width_vol = net(tsdf_vol, pos)
    return (qual_vol.cpu().squeeze().numpy(), rot_vol.cpu().squeeze().numpy(),
            width_vol.cpu().squeeze().numpy())


def predict_batch(tsdf_batch, pos, net):
    """Batched device-resident variant: tsdf_batch (B,40,40,40) cuda tensor, pos (1|B,N,3) -> device tensors."""
    from .convonet import _lattice_of
    B = tsdf_batch.shape[0]
    if pos.shape[0] == 1 and B > 1 and _lattice_of(pos) is None:
        pos = pos.expand(B, -1, -1).contiguous()          # a registered lattice is shared as-is
    with torch.no_grad():
        return net(tsdf_batch, pos)
