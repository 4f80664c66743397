"""Diagnostic: s_memtime timeline of workgroup 0's first round in the fused bf16 decoder backward (dect_kernel<true>); needs a
-DGIGA_TRACE build of the library passed as GIGA_DIAG_LIB (make -C giga_amd/csrc trace).   python tools/gpu_dect_trace.py [M]"""
import ctypes, os, sys
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).train().set_train_precision("bf16")
B = 32
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(0, B, M, stream=3)).to(dev)
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_dect_trace
dbg.argtypes = [ctypes.c_void_p]
names = {0: "round start", 1: "gather done", 2: "image landed", 3: "forward done", 4: "zone free", 5: "step5 tiles", 6: "barrier 5", 22: "last barrier", 23: "round end"}
for b in range(5):
    names[7 + 3 * b] = f"blk{4 - b} chain"; names[8 + 3 * b] = f"blk{4 - b} barrier"; names[9 + 3 * b] = f"blk{4 - b} wgrad"
for _ in range(3):
    net.zero_grad(set_to_none=True)
    sum(o.sum() for o in net(x, pos, p_tsdf=pos_occ)).backward()
torch.cuda.synchronize()
buf = np.zeros((2, 4, 40), np.int64)
dbg(buf.ctypes.data_as(ctypes.c_void_p))
for which, what in ((0, "grasp heads: head 0, one tile"), (1, f"occupancy head: workgroup 0, first of its rounds (M = {M})")):
    b = buf[which]
    t0 = b[:, 0][b[:, 0] > 0].min()
    print(what, "-- clocks since the first wave's round start; columns = waves")
    for k in range(40):
        if (b[:, k] == 0).all():
            continue
        print(f"{names.get(k, str(k)):14s}", " ".join(f"{int(v - t0) if v else -1:7d}" for v in b[:, k]))
