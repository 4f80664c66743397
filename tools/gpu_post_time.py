"""Timing of the device grasp post-processing and of the whole planner call (network + post-processing)."""
import time
import numpy as np
import torch
from giga_amd import synth
from giga_amd.detection import VGNImplicit, grasp_select
from giga_amd.networks import get_network
from giga_amd.weights import make_state_dict

dev = torch.device("cuda:0")
R = 40
for B in (1, 32):
    vols = [synth.post_volumes(s, R) for s in range(B)]
    st = lambda i: torch.from_numpy(np.stack([v[i] for v in vols])).to(dev)
    tsdf, qual, rot, width = st(0), st(1).reshape(B, -1), st(2).reshape(B, -1, 4), st(3).reshape(B, -1)
    for _ in range(3):
        grasp_select(tsdf, qual, rot, width, out_th=0.1, threshold=0.8)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20):
        sel = grasp_select(tsdf, qual, rot, width, out_th=0.1, threshold=0.8)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print(f"grasp_select B={B}: {dt*1e3:.3f} ms/call ({dt/B*1e6:.1f} us/scene), grasps/scene {np.mean([len(s['score']) for s in sel]):.1f}")

net = get_network("giga").to(dev)
net.load_state_dict(make_state_dict(7))
net.eval()
import itertools
for prec, graph in itertools.product(("fp32", "fp16"), (False, True)):
    net.set_precision(prec)
    planner = VGNImplicit(None, "giga", net=net, force_detection=True, qual_th=0.6, out_th=0.1, best=True, use_graph=graph)
    class S: pass
    s = S(); s.tsdf = synth.tsdf_batch(0, 1, realistic=True)
    for _ in range(3):
        planner(s)
    t0 = time.time()
    for _ in range(20):
        g, sc, toc = planner(s)
    dt = (time.time() - t0) / 20
    print(f"VGNImplicit.__call__ {prec} graph={graph}: {dt*1e3:.3f} ms/plan, {len(g)} grasps, best score {sc[0]:.6f}")
    tb = torch.from_numpy(synth.tsdf_batch(0, 32, realistic=True)).to(dev)
    for _ in range(3):
        planner.plan_batch(tb)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        planner.plan_batch(tb)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"plan_batch B=32 {prec} graph={graph}: {dt*1e3:.3f} ms ({32/dt:.0f} scenes/s)")
