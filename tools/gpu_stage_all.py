"""HIP-event time of EVERY encoder stage for each precision (median of 9).   PYTHONPATH=. python tools/gpu_stage_all.py [B ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

from giga_amd import _capi, networks, synth, weights

NAMES = ["convin_project", "plane_finalize", "down0.conv1", "down0.conv2+pool", "down1.conv1", "down1.conv2+pool", "down2.conv1",
         "down2.conv2", "up0.upconv", "up0.conv1", "up0.conv2", "up1.upconv", "up1.conv1", "up1.conv2", "conv_final"]
dev = torch.device("cuda:0")
net = networks.get_network("giga")
net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval()
L = _capi.lib()
ev = (L.giga_event_create(), L.giga_event_create())
ms = ctypes.c_float()
for B in [int(a) for a in sys.argv[1:]] or [32, 1]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    rows = {}
    PRECS = os.environ.get("GIGA_PRECS", "fp32,fp16,fp16x3").split(",")
    for prec in PRECS:
        blob = net.packed_blob(dev)
        with torch.no_grad():
            for _ in range(3):
                net.encoder.encode_nhwc(x, blob=blob, precision=prec)
            ts = []
            for st in range(15):
                t = []
                for _ in range(9):
                    net.encoder.encode_nhwc(x, blob=blob, precision=prec, probe=(st, ev[0], ev[1]))
                    _capi.check(L.giga_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)), "event")
                    t.append(ms.value)
                ts.append(float(np.median(t)) * 1e3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = []
            for _ in range(9):
                e0.record(); net.encoder.encode_nhwc(x, blob=blob, precision=prec); e1.record(); torch.cuda.synchronize()
                tot.append(e0.elapsed_time(e1) * 1e3)
        rows[prec] = (ts, float(np.median(tot)))
    print(f"B={B}: stage us    " + "".join(f"{p:>9s}" for p in PRECS))
    for i, n in enumerate(NAMES):
        print(f"  {n:20s}" + "".join(f" {rows[p][0][i]:8.1f}" for p in PRECS))
    print(f"  {'sum of stages':20s}" + "".join(f" {sum(rows[p][0]):8.1f}" for p in PRECS))
    print(f"  {'whole encoder':20s}" + "".join(f" {rows[p][1]:8.1f}" for p in PRECS))
