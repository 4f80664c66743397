"""Host-side (Python) cost of one planner call / one forward: cProfile of 300 calls on the GPU box.
Usage: PYTHONPATH=. python tools/gpu_host_profile.py"""
import cProfile, pstats, time
import numpy as np, torch
from giga_amd import networks, synth, weights
from giga_amd.detection import VGNImplicit

dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
net.set_precision("fp16")
class S: pass
st = S(); st.tsdf = synth.tsdf_batch(0, 1, realistic=True)
for graph in (False, True):
    pl = VGNImplicit(None, "giga", net=net, force_detection=True, qual_th=0.6, out_th=0.1, best=True, use_graph=graph)
    for _ in range(20):
        pl(st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(300):
        pl(st)
    print(f"graph={graph}: {(time.perf_counter() - t) / 300 * 1e3:.3f} ms/plan")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        pl(st)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
