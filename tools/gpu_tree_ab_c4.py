"""c4 step at 32 / 8 scenes and the single-scene planner replay in fp16 and fp16x3 through bench.py's own legs (oracle-checked);
use like tools/gpu_tree_ab_c2.py."""
import sys
sys.argv = ["bench.py"]
import torch
import bench
from giga_amd import _capi, networks, synth, weights
from giga_amd.convonet import decode_heads
import giga_amd
print("lib from", giga_amd.__file__, flush=True)
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
L = _capi.lib()
for prec in ("fp16", "fp16x3"):
    r = [bench.bench_c4(net, dev, L, _capi, synth, decode_heads, prec)["ms_per_step"] for _ in range(3)]
    g = [bench.bench_c4_graph(net, dev, synth, decode_heads, prec)["ms_per_call"] for _ in range(2)]
    r8 = [bench.bench_c4(net, dev, L, _capi, synth, decode_heads, prec, Bc=8, steps=10)["ms_per_step"] for _ in range(2)]
    print(prec, "c4@32", " ".join(f"{v:.4f}" for v in r), " c4@8", " ".join(f"{v:.4f}" for v in r8), " planner replay", " ".join(f"{v:.4f}" for v in g), flush=True)
