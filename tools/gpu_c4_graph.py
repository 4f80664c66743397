"""The single-scene (and small-batch) c4 network call as one hipGraph replay (bench.py's bench_c4_graph).
    PYTHONPATH=. python tools/gpu_c4_graph.py [scenes ...]"""
import sys
scenes = [int(a) for a in sys.argv[1:]] or [1]
sys.argv = ["bench.py"]
import torch
import bench
from giga_amd import networks, synth, weights
from giga_amd.convonet import decode_heads
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
for prec in ("fp16", "fp16x3", "fp32"):
    for B in scenes:
        for mode in ("layers", False):
            net.set_persistent_unet(mode)
            r = bench.bench_c4_graph(net, dev, synth, decode_heads, prec, Bc=B)
            print(f"{prec:7s} scenes {B:3d} unet={'per-layer' if mode else 'default  '}: {r['ms_per_call']*1e3:8.1f} us per replayed call   err {r['checked_vs_oracle']['max_abs_err']}", flush=True)
net.set_persistent_unet(False)
