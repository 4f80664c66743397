"""Diagnostic: issue timeline (s_memtime clocks) of workgroup 100 of the head-resident f16 decoder on ONE scene's 64 000
   lattice queries; needs a -DGIGA_TRACE build of the library passed as GIGA_DIAG_LIB.   python tools/gpu_dec_trace.py [fp16|fp16x3]"""
import ctypes, os, sys
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision(prec)
B = int(os.environ.get("GIGA_DIAG_B", "1"))
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
lat = torch.from_numpy(synth.inference_lattice()).to(dev)
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_dec_trace
dbg.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        net(x, lat)
torch.cuda.synchronize()
buf = np.zeros((16, 16), np.int64)
dbg(buf.ctypes.data_as(ctypes.c_void_p))
t0 = buf[:, 0][buf[:, 0] > 0].min()
names = {0: "entry", 1: "dma out", 15: "exit"}
for it in range(4):
    names[2 + 3 * it] = f"r{it} feat"; names[3 + 3 * it] = f"r{it} go"; names[4 + 3 * it] = f"r{it} done"
if B >= 4:      # decoder_lat_kernel's points
    names = {0: "entry", 1: "dma out", 2: "lines0", 3: "stored", 4: "barrier", 5: "tile1", 6: "tile2", 7: "tile3", 8: "slab0 tiles", 9: "lines1",
             10: "slab0 end", 11: "slab1 tiles", 12: "lines2", 13: "slab1 end"}
print("clocks since the first wave's entry; columns = waves")
for k in range(16):
    if (buf[:, k] == 0).all():
        continue
    print(f"{names.get(k, str(k)):9s}", " ".join(f"{int(v - t0) if v else -1:6d}" for v in buf[:12, k]))
