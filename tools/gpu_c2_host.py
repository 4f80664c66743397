"""Host-side cost of one c2 step (the Python + launch path of net(x, pos, p_tsdf=...)): wall time per call with the GPU kept
busy (calls return before the GPU finishes), and a cProfile of 300 calls.   PYTHONPATH=. python tools/gpu_c2_host.py"""
import cProfile
import pstats
import time
import torch
from giga_amd import networks, synth, weights
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
B = 32
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
occ = torch.from_numpy(synth.query_points(0, B, 2048, stream=3)).to(dev)
with torch.no_grad():
    for _ in range(20):
        net(x, pos, p_tsdf=occ)
    torch.cuda.synchronize()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); net(x, pos, p_tsdf=occ); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts.sort()
    print(f"host time per call: median {ts[50]*1e6:.1f} us, min {ts[0]*1e6:.1f}, p90 {ts[90]*1e6:.1f}")
    # time from an idle device to the completion of ONE call
    one = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); net(x, pos, p_tsdf=occ); torch.cuda.synchronize(); one.append(time.perf_counter() - t0)
    one.sort(); print(f"one call on an idle device, call + synchronize: median {one[10]*1e6:.1f} us, min {one[0]*1e6:.1f}")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        net(x, pos, p_tsdf=occ)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
