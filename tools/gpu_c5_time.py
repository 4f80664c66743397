"""Wall-clock time of the c5 training step (B = 32, 1 grasp + 2048 occupancy queries, flat parameter + FlatAdam) per training
precision, interleaved in one process (box drift cancels), with the library's launches per step.   python tools/gpu_c5_time.py [precisions...]"""
import sys
import time
import numpy as np
import torch
from giga_amd import _capi, networks, synth, weights
from giga_amd.optim import FlatAdam
from giga_amd.training import giga_loss

precs = sys.argv[1:] or ["fp32", "bf16_convs", "bf16"]
dev = torch.device("cuda:0")
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(2000, B)).to(dev); pos = torch.from_numpy(synth.query_points(2000, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(2000, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(2000, B, M))
nets = {}
for p in precs:
    net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train().set_train_precision(p)
    nets[p] = (net, FlatAdam(net.flatten_parameters(), lr=2e-4))


def step(p):
    net, opt = nets[p]
    opt.zero_grad(set_to_none=True)
    loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    loss.backward(); opt.step()
    return loss


res = {p: [] for p in precs}
L = _capi.lib()
for rep in range(6):
    for p in precs:
        for _ in range(5):
            step(p)
        torch.cuda.synchronize()
        n0 = L.giga_launch_count(); t0 = time.perf_counter()
        for _ in range(50):
            step(p)
        torch.cuda.synchronize()
        res[p].append((time.perf_counter() - t0) / 50 * 1e3)
        launches = (L.giga_launch_count() - n0) / 50
        if rep == 0:
            print(p, "library launches per step", launches)
for p in precs:
    print(f"{p:11s} ms per step: min {min(res[p]):.4f} median {float(np.median(res[p])):.4f}  all {[round(v, 4) for v in res[p]]}")
