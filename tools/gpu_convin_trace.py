"""Diagnostic: issue timeline (s_memtime, 100 MHz ticks) of one workgroup of convin_project_kernel<float,5>; needs a -DGIGA_TRACE
   build of the library passed as GIGA_DIAG_LIB.   python tools/gpu_convin_trace.py"""
import ctypes, os
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision(os.environ.get("GIGA_DIAG_PREC", "fp32"))
B = int(os.environ.get("GIGA_DIAG_B", "32"))
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_convin_trace
dbg.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        net.encoder.encode_nhwc(x)
torch.cuda.synchronize()
buf = np.zeros((8, 64), np.int64)
dbg(buf.ctypes.data_as(ctypes.c_void_p))
t0 = buf[:, 0][buf[:, 0] > 0].min()
names = {50: "loads out", 0: "entry", 1: "staged", 2: "barrier", 40: "loop end", 41: "red0", 42: "red1", 43: "red2", 63: "exit"}
for sx in range(5):
    for zg in range(5):
        names[3 + 6 * sx + zg] = f"s{sx} zg{zg}"
    names[3 + 6 * sx + 5] = f"s{sx} xy"
print("ticks of the 100 MHz s_memtime counter (10 ns) since the first wave's entry; columns = waves 0..7")
for k in range(64):
    if (buf[:, k] == 0).all():
        continue
    print(f"{names.get(k, str(k)):9s}", " ".join(f"{int(v - t0) if v else -1:6d}" for v in buf[:, k]))
