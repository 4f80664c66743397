"""conv16 against conv32 (the U-Net kernels of the f16-class modes) in ONE process, interleaved A B A B per batch size, so that drifts
of the box (clocks, temperature) hit both alike: encoder time and whole lattice call, each call synchronised (the latency regime).
    PYTHONPATH=. python tools/gpu_unet_ab.py [B ...]        GIGA_PRECS=fp16,fp16x3"""
import os
import sys

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.detection import predict_batch, query_lattice

dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
lat = query_lattice(40, dev)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    for prec in os.environ.get("GIGA_PRECS", "fp16,fp16x3").split(","):
        net.set_precision(prec)
        blob = net.packed_blob(dev)
        res = {"conv16": [], "conv32": []}
        with torch.no_grad():
            for rnd in range(3):
                for kernel in ("conv16", "conv32"):
                    net.set_unet_kernel(kernel)
                    enc = timed(lambda: net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True), 20)
                    call = timed(lambda: predict_batch(x, lat, net), 10)
                    res[kernel].append((enc, call))
        e16, c16 = np.median([r[0] for r in res["conv16"]]), np.median([r[1] for r in res["conv16"]])
        e32, c32 = np.median([r[0] for r in res["conv32"]]), np.median([r[1] for r in res["conv32"]])
        print(f"B={B:4d} {prec:7s} encoder us conv16 {e16:7.1f} conv32 {e32:7.1f} ({(e32 / e16 - 1) * 100:+5.1f} %)   lattice call us conv16 {c16:8.1f} conv32 {c32:8.1f} ({(c32 / c16 - 1) * 100:+5.1f} %)"
              f"   rounds conv16 {' '.join(f'{r[0]:.1f}' for r in res['conv16'])} | conv32 {' '.join(f'{r[0]:.1f}' for r in res['conv32'])}", flush=True)
net.set_unet_kernel("auto"); net.set_precision("fp32")
