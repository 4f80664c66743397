"""c2 step and encoder-only time in every inference precision, three repeats each.  Meant for A/B of two TREES on one box in one gpurun call
(regressions that no test sees -- round 5: 36 us lost in the fp16x3 encoder to a launch bound):
    git worktree add -f _old <commit> && (cd _old && python -m giga_amd.build) && cp tools/gpu_tree_ab_*.py _old/tools/
    gpurun -- 'PYTHONPATH=. python tools/gpu_tree_ab_c2.py; cd _old && PYTHONPATH=. python tools/gpu_tree_ab_c2.py'"""
import sys, time, torch, numpy as np
from giga_amd import networks, synth, weights
import giga_amd
print("lib from", giga_amd.__file__, flush=True)
dev = torch.device("cuda:0")
B, M = 32, 2048
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
occ = torch.from_numpy(synth.query_points(0, B, M, stream=3)).to(dev)
def t(fn, steps=100):
    with torch.no_grad():
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for prec in ("fp16x3", "fp16", "fp32"):
    net.set_precision(prec)
    blob = net.packed_blob(dev)
    full = [t(lambda: net(x, pos, p_tsdf=occ)) for _ in range(3)]
    enc = [t(lambda: net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)) for _ in range(3)]
    print(prec, "c2 step", " ".join(f"{v:.4f}" for v in full), " encoder only", " ".join(f"{v:.4f}" for v in enc), flush=True)
