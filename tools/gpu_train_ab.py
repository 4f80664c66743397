"""Step time of the c5-shaped training step (bench.py's bench_train legs with FlatAdam), repeated; GIGA_DIAG_LIB selects another
build of the library for an A/B.    PYTHONPATH=. python tools/gpu_train_ab.py [reps]"""
import os
import sys
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.argv = ["bench.py"]
import torch
import bench
from giga_amd import _capi, networks, synth, weights
if os.environ.get("GIGA_DIAG_LIB"):
    _capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
dev = torch.device("cuda:0")
for prec in os.environ.get("GIGA_TRAIN_PRECS", "bf16,fp32").split(","):
    for rep in range(reps):
        net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev)
        r = bench.bench_train(net, dev, synth, 32, 2048, steps=40, precision=prec, flat=True, giga_adam=True)
        print(f"train {prec:5s}: step {r['ms_per_step']*1e3:8.1f} us  median {r['step_ms_median']*1e3:8.1f}  final loss {r['final_loss']:.6f}", flush=True)
