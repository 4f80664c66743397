"""Where a Winograd layer goes wrong: per-layer errors and the error pattern of the first layer (A0) by pixel parity / tile / channel."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.nn.functional as F
from giga_amd import _capi, networks, synth, weights
from test_gpu_wino import layer_errors, NAMES, CH, HW

sd = weights.make_state_dict(7)
for form in ("layers", True):
    for Bs in (2, 32):
        errs, path, _ = layer_errors(sd, Bs, form)
        print(form, Bs, path, {k: "%.2e" % v for k, v in errs.items()})
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).eval().set_precision("fp32").set_persistent_unet("layers")
Bs = 2
x = torch.from_numpy(synth.tsdf_batch(40, Bs))
with torch.no_grad():
    net.encode_inputs(x.to(dev))
torch.cuda.synchronize()
ws = net.encoder._ws.snapshot()[-1]
off = (ctypes.c_size_t * 17)()
_capi.lib().giga_encoder_workspace_layout(Bs, 0, off)
def stage(nm):
    n = 3 * Bs * HW[nm] * HW[nm] * CH[nm]
    o = off[NAMES.index(nm)]
    return ws[o:o + 4 * n].view(torch.float32).view(3 * Bs, HW[nm], HW[nm], CH[nm]).permute(0, 3, 1, 2).double().cpu()
want = F.relu(F.conv2d(stage("P0"), sd["encoder.unet.down_convs.0.conv1.weight"].double(), sd["encoder.unet.down_convs.0.conv1.bias"].double(), padding=1))
got = stage("A0")
e = (got - want).abs().numpy()          # [img][c][y][x]
print("A0 err max", e.max(), "by image", e.max(axis=(1, 2, 3)))
print("by channel", np.round(e.max(axis=(0, 2, 3)), 3))
print("by (y%2, x%2)", [[float(e[:, :, a::2, b::2].max()) for b in range(2)] for a in range(2)])
print("by y", np.round(e[0].max(axis=(0, 2)), 3))
print("by x", np.round(e[0].max(axis=(0, 1)), 3))
print("got[0,0,:4,:4]", got[0, 0, :4, :4].numpy()); print("want", want[0, 0, :4, :4].numpy())
