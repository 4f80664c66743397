"""Every launch of one c5 training step (B = 32, 1 grasp + 2048 occupancy queries, flat parameter + FlatAdam), in launch order, timed
with HIP events on its own stream (giga_launch_probe; median of five steps).   python tools/gpu_c5_launches.py [precision]"""
import ctypes
import re
import sys
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
from giga_amd.optim import FlatAdam
from giga_amd.training import giga_loss

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(2000, B)).to(dev); pos = torch.from_numpy(synth.query_points(2000, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(2000, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(2000, B, M))
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train().set_train_precision(prec)
opt = FlatAdam(net.flatten_parameters(), lr=2e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    loss.backward(); opt.step()


L = _capi.lib()
for _ in range(10):
    step()
torch.cuda.synchronize()
n0 = L.giga_launch_count(); step(); torch.cuda.synchronize()
n = int(L.giga_launch_count() - n0)
ev0, ev1 = L.giga_event_create(), L.giga_event_create()
ms = ctypes.c_float()
tot = 0.0
print(f"# c5 step, train precision {prec}: {n} library launches (torch's own kernels -- fills, the loss glue -- are not listed)")
for i in range(1, n + 1):
    reps = []
    for _ in range(5):
        torch.cuda.synchronize()
        L.giga_launch_probe(L.giga_launch_count() + i, ev0, ev1)
        step()
        if L.giga_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)) == 0:
            reps.append(ms.value * 1e3)
    name = (L.giga_launch_probe_name() or b"").decode()
    name = re.sub(r"\(.*", "", name)[:110]
    t = float(np.median(reps)); tot += t
    print(f"{i:3d} {t:8.1f} us  {name}")
L.giga_launch_probe(0, None, None)
print(f"sum {tot:.1f} us (each bracket contains ~2-4 us of event overhead)")
