"""Diagnostic: issue timeline of workgroup 0 of one U-Net layer (GIGA_TRACE build: `make -C giga_amd/csrc trace`, GIGA_DIAG_LIB=giga_amd/lib/diag/libgiga_trace.so).
   python tools/gpu_conv_trace.py <layer 0..12> [...]"""
import ctypes, os, sys
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision(os.environ.get("GIGA_DIAG_PREC", "fp32"))
B = int(os.environ.get("GIGA_DIAG_B", "32"))
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
L = _capi.lib()
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_conv_trace
dbg.argtypes = [ctypes.c_int, ctypes.c_void_p]
for layer in [int(a) for a in sys.argv[1:]]:
    dbg(layer, None)
    with torch.no_grad():
        for _ in range(3):
            net.encoder.encode_nhwc(x)
    torch.cuda.synchronize()
    buf = np.zeros((12, 64), np.int64)   # (the Winograd stages of the fp32 path run 8 waves: columns 8..11 stay empty)
    dbg(layer, buf.ctypes.data_as(ctypes.c_void_p))
    t0 = buf[:, 0][buf[:, 0] > 0].min()
    print(f"=== layer {layer}: cycles since first wave entry; rows = events, columns = waves 0..11")
    names = {0: "entry", 1: "weights", 63: "exit"}
    for k in range(64):
        if (buf[:, k] == 0).all():
            continue
        nm = names.get(k, ("patch" if k % 2 == 0 else "mfma") + str((k - 2) // 2))
        print(f"{nm:9s}", " ".join(f"{int(v - t0) if v else -1:7d}" for v in buf[:, k]))
    buf[:] = 0
