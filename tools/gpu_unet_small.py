"""Small-batch U-Net: per-layer launches against the per-image persistent launch (GIGA_LAYERWISE_UNET / default), whole encoder
and whole network call on the inference lattice, with an equality check of the planes.
    PYTHONPATH=. python tools/gpu_unet_small.py [B ...]"""
import os
import sys

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.detection import predict_batch, query_lattice

dev = torch.device("cuda:0")
net = networks.get_network("giga")
net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval()
lat = query_lattice(40, dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 10]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    for prec in os.environ.get("GIGA_PRECS", "fp16,fp16x3,fp32").split(","):
        net.set_precision(prec)
        blob = net.packed_blob(dev)
        row, planes = {}, {}
        with torch.no_grad():
            for mode in ("layers", False, True):
                net.set_persistent_unet(mode)
                enc = lambda: net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)
                row[("enc", mode)] = timed(enc)
                row[("call", mode)] = timed(lambda: predict_batch(x, lat, net), 15)
                planes[mode] = enc()[0].float().clone()
        d = max((planes["layers"] - planes[m]).abs().max().item() for m in (False, True))
        print(f"B={B:3d} {prec:7s} encoder us: layers {row[('enc', 'layers')]:7.1f} default {row[('enc', False)]:7.1f} opt-in persistent {row[('enc', True)]:7.1f}   "
              f"network call us: {row[('call', 'layers')]:7.1f} / {row[('call', False)]:7.1f} / {row[('call', True)]:7.1f}   max |planes diff| {d:.3g}", flush=True)
net.set_persistent_unet(False)
