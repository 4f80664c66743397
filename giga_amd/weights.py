"""Deterministic, version-stable synthetic weights for the GIGA network.

There is no network access for the published checkpoints (`data/models/*.pt`, reference
README.md:29), so parity tests and the benchmark use weights generated here from
`numpy.random.default_rng` (bit-stable across numpy versions), keyed by the *reference's own
state-dict names and shapes* (`vgn.networks.GIGA`, networks.py:91-115).

Differences from the reference's init on purpose:
  * `blocks.*.fc_1.weight` is NOT zero (reference layers.py:37 zero-inits it, which would leave
    half of every ResNet block untested);
  * all biases are non-zero (reference unet.py:213-216 zeroes conv biases).
Gains are He-style so activations stay O(1) through the 14 conv layers and 5 residual blocks,
which makes absolute-error tolerances meaningful.
"""
import zlib
from collections import OrderedDict

import numpy as np

C_DIM = 32
HIDDEN = 32
N_BLOCKS = 5
HEADS = ("decoder_qual", "decoder_rot", "decoder_width", "decoder_tsdf")
HEAD_OUT_DIM = {"decoder_qual": 1, "decoder_rot": 4, "decoder_width": 1, "decoder_tsdf": 1}


def giga_param_shapes(with_tsdf=True, heads=None):
    """Ordered {name: shape} exactly as `get_network('giga').state_dict()` lays it out."""
    shapes = OrderedDict()
    if heads is None:
        heads = HEADS if with_tsdf else HEADS[:3]
    for h in heads:
        for i in range(N_BLOCKS):
            shapes[f"{h}.fc_c.{i}.weight"] = (HIDDEN, 3 * C_DIM)
            shapes[f"{h}.fc_c.{i}.bias"] = (HIDDEN,)
        shapes[f"{h}.fc_p.weight"] = (HIDDEN, 3)
        shapes[f"{h}.fc_p.bias"] = (HIDDEN,)
        for i in range(N_BLOCKS):
            for fc in ("fc_0", "fc_1"):
                shapes[f"{h}.blocks.{i}.{fc}.weight"] = (HIDDEN, HIDDEN)
                shapes[f"{h}.blocks.{i}.{fc}.bias"] = (HIDDEN,)
        shapes[f"{h}.fc_out.weight"] = (HEAD_OUT_DIM[h], HIDDEN)
        shapes[f"{h}.fc_out.bias"] = (HEAD_OUT_DIM[h],)
    e = "encoder."
    shapes[e + "conv_in.weight"] = (C_DIM, 1, 3, 3, 3)
    shapes[e + "conv_in.bias"] = (C_DIM,)
    u = e + "unet."
    down = [(32, 32), (32, 64), (64, 128)]
    for i, (ci, co) in enumerate(down):
        shapes[u + f"down_convs.{i}.conv1.weight"] = (co, ci, 3, 3)
        shapes[u + f"down_convs.{i}.conv1.bias"] = (co,)
        shapes[u + f"down_convs.{i}.conv2.weight"] = (co, co, 3, 3)
        shapes[u + f"down_convs.{i}.conv2.bias"] = (co,)
    up = [(128, 64), (64, 32)]
    for i, (ci, co) in enumerate(up):
        shapes[u + f"up_convs.{i}.upconv.weight"] = (ci, co, 2, 2)
        shapes[u + f"up_convs.{i}.upconv.bias"] = (co,)
        shapes[u + f"up_convs.{i}.conv1.weight"] = (co, 2 * co, 3, 3)
        shapes[u + f"up_convs.{i}.conv1.bias"] = (co,)
        shapes[u + f"up_convs.{i}.conv2.weight"] = (co, co, 3, 3)
        shapes[u + f"up_convs.{i}.conv2.bias"] = (co,)
    shapes[u + "conv_final.weight"] = (C_DIM, C_DIM, 1, 1)
    shapes[u + "conv_final.bias"] = (C_DIM,)
    return shapes


def _fan_in(name, shape):
    if name.endswith("upconv.weight"):  # ConvTranspose2d (Cin, Cout, 2, 2): each output sees Cin taps
        return shape[0]
    n = 1
    for s in shape[1:]:
        n *= s
    return n


def make_state_dict_numpy(seed=0, with_tsdf=True, heads=None):
    """{name: float32 ndarray}.  Each tensor has its own stream keyed by crc32(name) ^ seed."""
    out = OrderedDict()
    for name, shape in giga_param_shapes(with_tsdf, heads).items():
        rng = np.random.default_rng([zlib.crc32(name.encode()) & 0xFFFFFFFF, seed & 0xFFFFFFFF])
        if name.endswith("bias"):
            w = rng.uniform(-0.1, 0.1, size=shape)
        else:
            gain = 1.0
            if ".fc_1." in name:
                gain = 0.5      # residual branch: keep the stream from blowing up
            elif ".fc_c." in name:
                gain = 0.5
            elif name.endswith("fc_out.weight") or "conv_final" in name:
                gain = 0.7      # no ReLU follows
            a = gain * np.sqrt(6.0 / _fan_in(name, shape))
            w = rng.uniform(-a, a, size=shape)
        out[name] = w.astype(np.float32)
    return out


def make_state_dict(seed=0, with_tsdf=True, device="cpu", heads=None):
    import torch

    return OrderedDict((k, torch.from_numpy(v).to(device))
                       for k, v in make_state_dict_numpy(seed, with_tsdf, heads).items())


def num_params(with_tsdf=True):
    return sum(int(np.prod(s)) for s in giga_param_shapes(with_tsdf).values())
