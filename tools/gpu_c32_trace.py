"""Diagnostic: s_memtime timeline of the conv32 persistent U-Net kernel (member 0 of the group that holds image 0), per layer and wave:
entry of the layer, staging issued, staging barrier, tiles done, stores acknowledged + workgroup barrier, next weights requested,
group barrier released.  Needs the -DGIGA_TRACE build as GIGA_DIAG_LIB.   python tools/gpu_c32_trace.py [fp16|fp16x3|bf16]"""
import ctypes, os, sys
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision(prec)
B = int(os.environ.get("GIGA_DIAG_B", "32"))
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
dbg = ctypes.CDLL(_capi.LIB_PATH).giga_debug_c32_trace
dbg.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(5):
        net.encoder.encode_nhwc(x, fold_final=True)
torch.cuda.synchronize()
buf = np.zeros((13, 8, 8), np.int64)
dbg(buf.ctypes.data_as(ctypes.c_void_p))
t0 = buf[0, :, 0].min()
print(f"B={B} {prec}: clocks since layer 0 entry; per layer (wave 0 / slowest wave): entry | +stage issued | +stage barrier | +tiles | +stores acked | next weights requested | group barrier released")
prev = 0
for l in range(12):
    r = buf[l]
    e = r[:, 0].min()
    cols = [int(r[:, i].max() - e) if (r[:, i] > 0).all() else -1 for i in range(1, 5)]
    nxt = buf[l + 1] if l + 1 < 13 else None
    w5 = int(nxt[:, 5].max() - e) if nxt is not None and (nxt[:, 5] > 0).all() else -1
    w6 = int(nxt[:, 6].max() - e) if nxt is not None and (nxt[:, 6] > 0).all() else -1
    print(f"L{l:2d} entry {int(e - t0):8d} | stage issued {cols[0]:6d} | barrier {cols[1]:6d} | tiles {cols[2]:6d} | acked {cols[3]:6d} | weights {w5:6d} | released {w6:6d}"
          f"   (tiles per wave: {' '.join(str(int(r[w, 3] - r[w, 2])) for w in range(8))})")
