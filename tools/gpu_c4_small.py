"""c4 decoder time / fraction of the f16 MFMA peak at small scene counts (bench.py's bench_c4 leg).
   PYTHONPATH=. python tools/gpu_c4_small.py [scenes ...]"""
import sys
scenes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
sys.argv = ["bench.py"]
import torch
import bench
from giga_amd import _capi, networks, synth, weights
from giga_amd.convonet import decode_heads
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
L = _capi.lib()
for prec in ("fp16", "fp16x3"):
    for B in scenes:
        r = bench.bench_c4(net, dev, L, _capi, synth, decode_heads, prec, Bc=B, steps=20)
        print(f"{prec:7s} scenes {B:3d}: step {r['ms_per_step']*1e3:8.1f} us  decoder {r['roofline']['avg_launch_ms']*1e3:8.1f} us  "
              f"frac of f16 peak {r['roofline']['frac']:.4f}")
