"""Stage-by-stage comparison of the bf16 U-Net forward with the oracle evaluated on bf16-rounded operands.
    PYTHONPATH=. python tests/diag/gpu_bf16_stages.py"""
import ctypes

import torch

from giga_amd import _capi, networks, synth, weights
from oracle import giga_oracle as O

STAGES = ["P0", "A0", "S0", "Q0", "A1", "S1", "Q1", "A2", "S2", "U0", "A3", "A4", "U1", "A5", "A6"]
CH = {"P0": 32, "A0": 32, "S0": 32, "Q0": 32, "A1": 64, "S1": 64, "Q1": 64, "A2": 128, "S2": 128, "U0": 64, "A3": 64, "A4": 64, "U1": 32, "A5": 32, "A6": 32}
HW = {"P0": 40, "A0": 40, "S0": 40, "Q0": 20, "A1": 20, "S1": 20, "Q1": 10, "A2": 10, "S2": 10, "U0": 20, "A3": 20, "A4": 20, "U1": 40, "A5": 40, "A6": 40}
dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).eval()
B = 2
x = torch.from_numpy(synth.tsdf_batch(300, B))
bf = lambda t: t.bfloat16().float()  # noqa: E731
planes_in = O.project_planes(O.conv_in_relu(sd, x))
L = _capi.lib()
for prec, rnd in (("fp32", None), ("bf16", bf)):
    net.set_precision(prec)
    with torch.no_grad():
        net.encode_inputs(x.to(dev))
    ws = net.encoder._ws.snapshot()[-1]
    off = (ctypes.c_size_t * 17)()
    L.giga_encoder_workspace_layout(B, _capi.PRECISION[prec], off)
    ref = {k: O.unet_stages(sd, planes_in[k], rnd=rnd) for k in O.PLANES}
    print("==", prec)
    for i, st in enumerate(STAGES):
        n = 3 * B * HW[st] * HW[st] * CH[st]
        got = ws[off[i]:off[i] + 4 * n].view(torch.float32).view(3, B, HW[st], HW[st], CH[st]).permute(0, 1, 4, 2, 3).cpu()
        want = torch.stack([planes_in[k] if st == "P0" else ref[k][st] for k in O.PLANES])
        d = (got - want).abs()
        print(f"  {st}: max err {d.max().item():.3e}  mean err {d.mean().item():.3e}  |ref| max {want.abs().max().item():.3f}")
