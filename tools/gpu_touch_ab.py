"""Timings that depend on the persistent U-Net's output pre-read (GIGA_UNET_TOUCH, read once per process): the c2 step in fp32 / fp16 /
fp16x3 at 32 and 128 scenes, the encoder alone, and the c5 bf16 training step.  Run once per setting.   python tools/gpu_touch_ab.py"""
import os
import time
import numpy as np
import torch
from giga_amd import networks, synth, weights
from giga_amd.optim import FlatAdam
from giga_amd.training import giga_loss

dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()


def timed(fn, n=60):
    for _ in range(15):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return min(ts), float(np.median(ts))


print("GIGA_UNET_TOUCH =", os.environ.get("GIGA_UNET_TOUCH", "(default)"), " GIGA_C32_TOUCH =", os.environ.get("GIGA_C32_TOUCH", "(default)"),
      " GIGA_CONV32 =", os.environ.get("GIGA_CONV32", "(default)"))
for B in [int(v) for v in os.environ.get("GIGA_AB_BATCHES", "32,128").split(",")]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
    occ = torch.from_numpy(synth.query_points(0, B, 2048, stream=3)).to(dev)
    for prec in ("fp32", "fp16", "fp16x3"):
        net.set_precision(prec)
        with torch.no_grad():
            a = timed(lambda: net(x, pos, p_tsdf=occ))
            b = timed(lambda: net.encode_inputs(x))
        print(f"B={B:4d} {prec:7s} c2 step min {a[0]:.4f} median {a[1]:.4f} ms | encoder alone min {b[0]:.4f} median {b[1]:.4f} ms")
net.set_precision("fp32")
if os.environ.get("GIGA_AB_NO_TRAIN"):
    raise SystemExit(0)
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(2000, B)).to(dev); pos = torch.from_numpy(synth.query_points(2000, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(2000, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(2000, B, M))
for prec in ("bf16", "fp32"):
    n2 = networks.get_network("giga"); n2.load_state_dict(weights.make_state_dict(7)); n2 = n2.to(dev).train().set_train_precision(prec)
    opt = FlatAdam(n2.flatten_parameters(), lr=2e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = giga_loss(n2(x, pos, p_tsdf=pos_occ), y)
        loss.backward(); opt.step()
    a = timed(step, 40)
    print(f"c5 train step {prec}: min {a[0]:.4f} median {a[1]:.4f} ms")
