"""Diagnostic: per-layer arrival / release clocks (s_memtime) at the XCD barriers of the persistent U-Net kernel, for eight
   workgroups of XCD 0; needs a -DGIGA_TRACE build as GIGA_DIAG_LIB.   python tools/gpu_unet_trace.py [fp32|fp16|fp16x3]"""
import ctypes, os, sys
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision(prec)
B = int(os.environ.get("GIGA_DIAG_B", "32"))
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_mega_trace
dbg.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(5):
        net.encoder.encode_nhwc(x)
torch.cuda.synchronize()
buf = np.zeros((8, 32), np.int64)
dbg(buf.ctypes.data_as(ctypes.c_void_p))
t0 = buf[:, 2][buf[:, 2] > 0].min()
print("clocks; rows = barrier before layer l (arrive / release), columns = workgroups 0, 8, .., 56 of XCD 0")
prev = None
for l in range(1, 13):
    a, r = buf[:, 2 * l], buf[:, 2 * l + 1]
    if (a == 0).all():
        continue
    print(f"L{l:2d} arrive ", " ".join(f"{int(v - t0):7d}" for v in a), "  release", int(r.max() - t0), " wait of the last arriver", int(r.max() - a.max()))
