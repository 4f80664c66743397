"""Model registry: drop-in for the GIGA entries of the reference's `vgn.networks`
(/root/reference/src/vgn/networks.py:10-35, 65-169).  Same names, same hyper-parameter dicts,
same checkpoint convention (`torch.load(path)` -> `load_state_dict`, file stem `vgn_<name>_<n>`)."""
from pathlib import Path

import torch

from .convonet import get_model


def _cfg(decoder_tsdf, **extra):
    cfg = {
        "encoder": "voxel_simple_local",
        "encoder_kwargs": {
            "plane_type": ["xz", "xy", "yz"],
            "plane_resolution": 40,
            "unet": True,
            "unet_kwargs": {"depth": 3, "merge_mode": "concat", "start_filts": 32},
        },
        "decoder": "simple_local",
        "decoder_tsdf": decoder_tsdf,
        "decoder_kwargs": {"dim": 3, "sample_mode": "bilinear", "hidden_size": 32, "concat_feat": True},
        "padding": 0,
        "c_dim": 32,
    }
    cfg.update(extra)
    return cfg


def GIGAAff():        # networks.py:65-89
    return get_model(_cfg(False))


def GIGA():           # networks.py:91-115
    return get_model(_cfg(True))


def GIGAGeo():        # networks.py:117-142
    return get_model(_cfg(True, tsdf_only=True))


def GIGADetach():     # networks.py:144-169
    return get_model(_cfg(True, detach_tsdf=True))


def get_network(name):
    """networks.py:10-18.  ('vgn', the dense 3-D-conv baseline, is a different model: out of scope.)"""
    models = {"giga_aff": GIGAAff, "giga": GIGA, "giga_geo": GIGAGeo, "giga_detach": GIGADetach}
    key = name.lower()
    if key == "vgn":
        raise NotImplementedError("the VGN ConvNet baseline is not part of the GIGA hot path")
    return models[key]()


def load_network(path, device, model_type=None):
    """networks.py:21-35: build the network named by `model_type` (or by the file stem) and load
    the reference checkpoint (a plain state-dict)."""
    path = Path(path)
    model_name = "_".join(path.stem.split("_")[1:-1]) if model_type is None else model_type
    print(f"Loading [{model_type}] model from {path}")
    net = get_network(model_name).to(device)
    net.load_state_dict(torch.load(path, map_location=device))
    return net
