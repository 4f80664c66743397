"""Per-stage HIP-event timing of the encoder at several batch sizes (stage 0 = conv_in + projection).
    PYTHONPATH=. python tools/gpu_stage_time.py [stage ...]"""
import ctypes
import sys

import numpy as np
import torch

from giga_amd import _capi, networks, synth, weights

import os
if os.environ.get("GIGA_DIAG_LIB"):                      # diagnostic builds (ablation variants of a kernel)
    _capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
stages = [int(a) for a in sys.argv[1:]] or [0]
BS = [int(b) for b in os.environ.get("GIGA_DIAG_B", "8,16,32,64,128,256").split(",")]
dev = torch.device("cuda:0")
net = networks.get_network("giga")
net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision(os.environ.get("GIGA_DIAG_PREC", "fp32"))
L = _capi.lib()
ev = (L.giga_event_create(), L.giga_event_create())
ms = ctypes.c_float()
for B in BS:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
    with torch.no_grad():
        for _ in range(3):
            net(x, pos)
        for st in stages:
            t = []
            for _ in range(20):
                net(x, pos, _probe=(st, ev[0], ev[1]))
                _capi.check(L.giga_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)), "event")
                t.append(ms.value)
            print(f"B={B:4d} stage {st}: median {np.median(t)*1e3:8.1f} us  min {np.min(t)*1e3:8.1f} us  per scene {np.median(t)*1e3/B:6.2f} us")
