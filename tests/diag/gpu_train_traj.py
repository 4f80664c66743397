"""Loss trajectory of the c5-shaped training loop on the GPU (fused loss vs the torch helpers), with per-step timing.
    PYTHONPATH=. python tests/diag/gpu_train_traj.py [steps]"""
import sys
import time

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.training import giga_loss
from oracle import giga_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(2000, B)).to(dev)
pos = torch.from_numpy(synth.query_points(2000, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(2000, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(2000, B, M))
for kind in ("fused", "torch-helpers"):
    for fused_adam in (True, False):
        net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
        opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=fused_adam)
        losses, ts = [], []
        v0 = next(net.parameters())._version
        for i in range(steps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            out = net(x, pos, p_tsdf=pos_occ)
            loss, d = giga_loss(out, y) if kind == "fused" else O.train_loss(O.train_select(out), y)
            loss.backward(); opt.step()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            losses.append(float(loss.detach()))
        print(kind, "fused_adam" if fused_adam else "foreach_adam", "version bump/step", (next(net.parameters())._version - v0) / steps,
              "loss", " ".join(f"{l:.3f}" for l in losses[::max(1, steps // 12)]), f"| ms median {np.median(ts):.2f} max {np.max(ts):.2f}")
