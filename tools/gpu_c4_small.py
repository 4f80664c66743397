"""c4 decoder time / fraction of the f16 MFMA peak at small scene counts (bench.py's bench_c4 leg).
   PYTHONPATH=. python tools/gpu_c4_small.py [scenes ...]"""
import sys
scenes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
sys.argv = ["bench.py"]
import torch
import bench
from giga_amd import _capi, networks, synth, weights
from giga_amd.convonet import decode_heads
import os
if os.environ.get("GIGA_DIAG_LIB"):                  # A/B against another build of the library
    _capi.LIB_PATH = os.environ["GIGA_DIAG_LIB"]
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
L = _capi.lib()
modes = os.environ.get("GIGA_C4_MODES", "default").split(",")        # "layers" = one launch per U-Net layer, "default"
for prec in [q for q in os.environ.get("GIGA_C4_PRECS", "fp16,fp16x3").split(",") if q and q != "none"]:
    for B in scenes:
        for rep in range(int(os.environ.get("GIGA_C4_REPS", "1"))):
            for mode in modes:
                net.set_persistent_unet("layers" if mode == "layers" else False)
                r = bench.bench_c4(net, dev, L, _capi, synth, decode_heads, prec, Bc=B, steps=20)
                print(f"{prec:7s} scenes {B:3d} {mode:8s}: step {r['ms_per_step']*1e3:8.1f} us  decoder {r['roofline']['avg_launch_ms']*1e3:8.1f} us  "
                      f"frac of f16 peak {r['roofline']['frac']:.4f}", flush=True)
net.set_persistent_unet(False)

if os.environ.get("GIGA_C4_C2"):                     # the headline workload (c2, 32 scenes) in the given precisions
    x = torch.from_numpy(synth.tsdf_batch(0, 32)).to(dev)
    pos = torch.from_numpy(synth.query_points(0, 32, 1, stream=2)).to(dev)
    occ = torch.from_numpy(synth.query_points(0, 32, 2048, stream=3)).to(dev)
    for prec in os.environ["GIGA_C4_C2"].split(","):
        for rep in range(3):
            r = bench.bench_c2_mode(net, x, pos, occ, prec, steps=30)
            print(f"c2 {prec:7s}: step {r['ms_per_step']*1e3:8.1f} us", flush=True)
    net.set_precision("fp32")
