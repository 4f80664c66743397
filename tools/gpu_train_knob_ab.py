"""In-process A/B of a per-call environment knob on the c5-shaped training step (bf16 / fp32, FlatAdam): the values are interleaved
v0 v1 v2 v0 v1 v2 ..., every leg = 30 steps on a fresh network with the same seed; prints the median step and the final loss.
    PYTHONPATH=. python tools/gpu_train_knob_ab.py GIGA_WGRAD_GX_DIV 1 2 4"""
import os
import sys
knob, values = sys.argv[1], sys.argv[2:]
sys.argv = ["bench.py"]
import numpy as np
import torch
import bench
from giga_amd import networks, synth, weights
dev = torch.device("cuda:0")
for prec in os.environ.get("GIGA_TRAIN_PRECS", "bf16,fp32").split(","):
    res = {v: [] for v in values}
    loss = {}
    for rnd in range(3):
        for v in values:
            os.environ[knob] = v
            net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev)
            r = bench.bench_train(net, dev, synth, 32, 2048, steps=30, precision=prec, flat=True, giga_adam=True)
            res[v].append(r["step_ms_median"] * 1e3); loss[v] = r["final_loss"]
    for v in values:
        print(f"train {prec:5s} {knob}={v}: median step {np.median(res[v]):8.1f} us   rounds {' '.join(f'{x:.1f}' for x in res[v])}   final loss {loss[v]:.6f}", flush=True)
os.environ.pop(knob, None)
