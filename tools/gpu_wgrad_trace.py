"""Diagnostic: issue timeline of workgroup (0,0) of the LAST conv3_wgrad_kernel launch of a backward pass (layer 0:
32->32 at 40x40), GIGA_TRACE build."""
import ctypes, os
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
from giga_amd.training import giga_loss
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
net.set_train_precision(os.environ.get("GIGA_DIAG_TRAIN_PREC", "fp32"))    # bf16: conv3_wgrad_bf16_kernel (same trace slots)
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(0, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(0, B, M))
for _ in range(2):
    loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y); loss.backward()
torch.cuda.synchronize()
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_wgrad3_trace
dbg.argtypes = [ctypes.c_void_p]
buf = np.zeros((8, 32), np.int64)
dbg(buf.ctypes.data_as(ctypes.c_void_p))
t0 = buf[:, 0].min()
for kx in range(32):
    if (buf[:, kx] == 0).all(): continue
    print(f"ev{kx:2d}", " ".join(f"{int(v - t0):8d}" for v in buf[:, kx]))
