"""Diagnostic: per-wave issue timeline of convin_project workgroup (0,0) (needs the CI_TRACE build of the library:
GIGA_DIAG_LIB=giga_amd/lib/abl_trace.so)."""
import ctypes, os
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
_capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
net = net.to(dev).eval().set_precision("fp32")
B = int(os.environ.get("GIGA_DIAG_B", "32"))
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
L = _capi.lib()
lay = (ctypes.c_size_t * 32)()
L.giga_encoder_workspace_layout(B, 0, lay)
names = "P0 A0 S0 Q0 A1 S1 Q1 A2 S2 U0 A3 A4 U1 A5 A6 YZ XZ total".split()
off = dict(zip(names, list(lay)))
with torch.no_grad():
    for _ in range(3):
        net.encoder.encode_nhwc(x)
    torch.cuda.synchronize()
ws = next(iter(net.encoder._ws.values()))
t = ws[off["YZ"]: off["YZ"] + 8 * 128 * 8].view(torch.int64).cpu().numpy().reshape(8, 128)
SX = 40 // max(d for d in (1, 2, 4, 5, 8, 10, 20, 40) if True and B * d >= 256 and all(B * e < 256 for e in (1, 2, 4, 5, 8, 10, 20, 40) if e < d)) if B < 256 else 40
t0 = t[:, 0].min()
lab = ["start"] + [f"{'mfma' if i % 2 == 0 else 'epi'}{i // 2}" for i in range(10)] + ["pre_bar", "post_bar", "post_red"]
for sx in range(min(SX, 3)):
    print(f"--- slice {sx}: cycles since kernel t0; columns = waves 0..7")
    for k, name in enumerate(lab):
        print(f"{name:9s}", " ".join(f"{int(v - t0):7d}" for v in t[:, sx * 16 + k]))
