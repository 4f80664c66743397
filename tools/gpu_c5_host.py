"""Is the c5 training step host-bound?  Enqueue time of a step on the host (no synchronisation inside the loop) against its wall time:
the host must stay ahead of the device for the stream to run dry-free.   python tools/gpu_c5_host.py [precision ...]"""
import sys
import time
import torch
from giga_amd import networks, synth, weights
from giga_amd.optim import FlatAdam
from giga_amd.training import giga_loss

dev = torch.device("cuda:0")
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(2000, B)).to(dev); pos = torch.from_numpy(synth.query_points(2000, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(2000, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(2000, B, M))
for prec in sys.argv[1:] or ["bf16", "fp32"]:
    net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train().set_train_precision(prec)
    opt = FlatAdam(net.flatten_parameters(), lr=2e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
        loss.backward(); opt.step()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = time.perf_counter() - t0            # the loop returns when the last step is ENQUEUED
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # one step at a time (the host's enqueue time with an idle device in front of it)
    hs = []
    for _ in range(30):
        torch.cuda.synchronize(); t1 = time.perf_counter(); step(); hs.append(time.perf_counter() - t1)
    hs.sort()
    print(f"{prec}: {n} steps enqueued in {t_host / n * 1e3:.4f} ms per step, finished after {t_all / n * 1e3:.4f} ms per step; "
          f"enqueue of ONE step on an idle device: median {hs[len(hs) // 2] * 1e3:.4f} ms", flush=True)
