"""Live check of the oracle against the reference imported in the build container.
Skipped wherever /root/reference is absent (e.g. the GPU box)."""
import numpy as np
import pytest
import torch

from giga_amd import synth, weights
from oracle import giga_oracle as O
from oracle import ref_bootstrap

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_bootstrap.reference_available(), reason="no /root/reference")]


@pytest.fixture(scope="module")
def ref_and_sd():
    sd = weights.make_state_dict(11)
    return ref_bootstrap.load_reference_giga(sd), sd


def test_state_dict_keys_match_reference(ref_and_sd):
    net, sd = ref_and_sd
    ref_sd = net.state_dict()
    assert list(ref_sd.keys()) == list(weights.giga_param_shapes().keys())
    for k, v in ref_sd.items():
        assert tuple(v.shape) == weights.giga_param_shapes()[k], k


def test_forward_matches_reference(ref_and_sd):
    net, sd = ref_and_sd
    x = torch.from_numpy(synth.tsdf_batch(20, 2, realistic=True))
    p = torch.from_numpy(synth.query_points(20, 2, 777, stream=5, half_width=0.55))
    with torch.no_grad():
        ref = net(x, p, p_tsdf=p)
        got = O.model_forward(sd, x, p, p_tsdf=p)
        ref_c = net.encode_inputs(x)
        got_c = O.encoder_forward(sd, x)
    for a, b in zip(ref, got):
        assert (a - b).abs().max().item() < 2e-5
    for k in O.PLANES:
        assert (ref_c[k] - got_c[k]).abs().max().item() < 2e-5


def test_scatter_mean_is_axis_mean(ref_and_sd):
    """The closed form of SURVEY 8a/a4: every plane cell receives exactly the 40 voxels along the
    projected axis, independent of the scatter stub's arithmetic."""
    from vgn.ConvONets.common import coordinate2index, normalize_coordinate
    lin = torch.linspace(-0.5, 0.5, 40)
    gx, gy, gz = torch.meshgrid(lin, lin, lin, indexing="ij")
    p = torch.stack((gx, gy, gz), -1).reshape(1, -1, 3)
    ijk = torch.stack(torch.meshgrid(*(torch.arange(40),) * 3, indexing="ij"), -1).reshape(-1, 3)
    for plane, (a0, a1) in (("xz", (0, 2)), ("xy", (0, 1)), ("yz", (1, 2))):
        idx = coordinate2index(normalize_coordinate(p.clone(), plane=plane, padding=0), 40)[0, 0]
        assert torch.equal(idx, ijk[:, a0] + 40 * ijk[:, a1])
        assert torch.equal(torch.bincount(idx, minlength=1600), torch.full((1600,), 40))
