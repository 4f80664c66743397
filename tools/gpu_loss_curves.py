"""fp32 vs bf16 training on the same data, same initial weights, same optimizer (does the bf16 step TRAIN like the
fp32 one, beyond per-tensor gradient parity?).  300 steps of scripts/train_giga.py:198-211 (forward, joint loss, backward, Adam 2e-4) on
a pool of 16 synthetic batches of 32 scenes (1 grasp query + 2048 occupancy queries each), cycled.  Targets are LEARNABLE from the input so
that the curve is not a flat entropy plateau: occupancy label = TSDF of the query's voxel > 0.5, grasp label = the same at the grasp
query; rotations / widths are synth.train_labels' (random: a noise floor common to both runs).
    PYTHONPATH=. python tools/gpu_loss_curves.py [steps]"""
import sys

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.optim import FlatAdam
from giga_amd.training import giga_loss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
B, M, POOL = 32, 2048, 16


def voxel_label(x, p):
    """x (B,40,40,40), p (B,N,3) in [-0.5, 0.5): 1.0 where the TSDF voxel that contains p is > 0.5"""
    idx = np.clip(((p + 0.5) * 40).astype(np.int64), 0, 39)
    b = np.arange(x.shape[0])[:, None]
    return (x[b, idx[..., 0], idx[..., 1], idx[..., 2]] > 0.5).astype(np.float32)


pool = []
for k in range(POOL):
    first = 5000 + k * B
    x = synth.tsdf_batch(first, B)
    pos = synth.query_points(first, B, 1, stream=2)
    pos_occ = synth.query_points(first, B, M, stream=3)
    label, rot, width, occ = synth.train_labels(first, B, M)
    label = voxel_label(x, pos).reshape(label.shape).astype(label.dtype)
    occ = voxel_label(x, pos_occ).reshape(occ.shape).astype(occ.dtype)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (x, pos, pos_occ, label, rot, width, occ)))

curves = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(0)
    net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7))
    net = net.to(dev).train().set_train_precision(prec)
    opt = FlatAdam(net.flatten_parameters(), lr=2e-4)
    losses, parts = [], []
    for i in range(steps):
        x, pos, pos_occ, *y = pool[i % POOL]
        opt.zero_grad(set_to_none=True)
        loss, d = giga_loss(net(x, pos, p_tsdf=pos_occ), tuple(y))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        parts.append({k: float(v) for k, v in d.items()} if isinstance(d, dict) else {})
    curves[prec] = (np.array(losses), parts)
    print(f"{prec}: first {losses[0]:.4f}  last {losses[-1]:.4f}")

a, b = curves["fp32"][0], curves["bf16"][0]
print("step   fp32(mean of 10)  bf16(mean of 10)  bf16/fp32-1")
for s in range(0, steps, 10):
    ma, mb = a[s:s + 10].mean(), b[s:s + 10].mean()
    print(f"{s:4d}   {ma:10.4f}        {mb:10.4f}        {mb / ma - 1:+.4f}")
tail = slice(steps - 50, steps)
print(f"last 50 steps: fp32 {a[tail].mean():.4f}  bf16 {b[tail].mean():.4f}  relative gap {b[tail].mean() / a[tail].mean() - 1:+.4f}; "
      f"loss drop fp32 {a[:10].mean() - a[tail].mean():.4f}  bf16 {b[:10].mean() - b[tail].mean():.4f}")
for k in (curves["fp32"][1][-1] or {}):
    print(f"  last-step term {k}: fp32 {curves['fp32'][1][-1][k]:.4f}  bf16 {curves['bf16'][1][-1][k]:.4f}")
