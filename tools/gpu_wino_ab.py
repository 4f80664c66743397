"""Winograd F(2x2, 3x3) vs the direct 3x3 convolutions of the fp32 U-Net (csrc/giga_wino.h): planes against each other (the comparison
with the CPU oracle lives with the tests: tests/test_gpu_wino.py, tests/diag/gpu_wino_diag.py), per-stage HIP-event times in both forms.
    PYTHONPATH=. python tools/gpu_wino_ab.py [B]"""
import ctypes
import sys

import numpy as np
import torch

from giga_amd import _capi, networks, synth, weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
net = networks.get_network("giga")
net.load_state_dict(sd)
net = net.to(dev).eval().set_precision("fp32")
L = _capi.lib()
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
res = {}
with torch.no_grad():
    for form in ("layers", False):
        net.set_persistent_unet(form)
        for kern in ("direct", "auto"):
            net.set_unet_kernel(kern)
            nhwc, nchw = net.encoder.encode_nhwc(x, want_nchw=True)
            torch.cuda.synchronize()
            res[(form, kern)] = nchw.float().cpu()
            print(f"launch form {form!r:9} kernel {kern:7} path flags {L.giga_encoder_last_path()}")
for k, v in res.items():
    d = (v - res[("layers", "direct")]).abs().max().item()
    print(f"{k}: max |planes - direct per-layer| = {d:.3e}  (planes max {v.abs().max().item():.3f})")
ev = (L.giga_event_create(), L.giga_event_create())
ms = ctypes.c_float()
pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
names = ["convin", "finalize", "d0c1", "d0c2+pool", "d1c1", "d1c2+pool", "d2c1", "d2c2", "up0.up", "up0.c1", "up0.c2", "up1.up", "up1.c1", "up1.c2",
         "final", "unet(persistent)"]
net.set_persistent_unet(False)
with torch.no_grad():
    for kern in ("direct", "auto"):
        net.set_unet_kernel(kern)
        for _ in range(5):
            net(x, pos)
        row = []
        for st in list(range(2, 14)) + [15]:
            t = []
            for _ in range(15):
                net(x, pos, _probe=(st, ev[0], ev[1]))
                _capi.check(L.giga_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)), "event")
                t.append(ms.value)
            row.append(np.median(t) * 1e3)
        print(f"B={B} {kern:7}: " + "  ".join(f"{names[st]} {v:6.1f}" for st, v in zip(list(range(2, 14)) + [15], row)), " us")
