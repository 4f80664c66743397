"""Where does a fed training step spend its time?  PYTHONPATH=. python tools/gpu_feed_trace.py"""
import tempfile
import time

import numpy as np
import torch

from giga_amd import dataset, networks, synth, weights
from giga_amd.feed import TSDFFeed
from giga_amd.training import giga_loss

B, M = 32, 2048
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as tmp:
    root, raw = tmp + "/data", tmp + "/raw"
    synth.write_training_set(root, raw, n_scenes=64, grasps_per_scene=24, occ_files=(2, 4), n_occ_points=20000, seed=9)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=M)
    net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)
    for nw in (0, 8):
        it = iter(dataset.GraspOccBatches(ds, B, seed=2, drop_last=True, workers=nw))
        t_host, t_pin, t_h2d, t_prep, t_step = [], [], [], [], []
        pins = None
        for k in range(40):
            t0 = time.perf_counter(); hb = next(it); t1 = time.perf_counter()
            leaves = [hb[0], *hb[1], hb[2], hb[3], hb[4]]
            leaves = [torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a for a in leaves]
            if pins is None:
                pins = [torch.empty(a.shape, dtype=a.dtype).pin_memory() for a in leaves]
            for p_, a in zip(pins, leaves):
                p_.copy_(a)
            t2 = time.perf_counter()
            d = [p_.to(dev, non_blocking=True) for p_ in pins]
            torch.cuda.synchronize(); t3 = time.perf_counter()
            x, pos, pocc, y = dataset.network_inputs((d[0], (d[1], d[2], d[3]), d[4], d[5], d[6]))
            torch.cuda.synchronize(); t4 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            loss, _ = giga_loss(net(x, pos, p_tsdf=pocc), y); loss.backward(); opt.step()
            torch.cuda.synchronize(); t5 = time.perf_counter()
            if k >= 5:
                t_host.append(t1 - t0); t_pin.append(t2 - t1); t_h2d.append(t3 - t2); t_prep.append(t4 - t3); t_step.append(t5 - t4)
        f = lambda v: f"{np.median(v) * 1e3:7.2f}"  # noqa: E731
        print(f"workers={nw}: next(host batch) {f(t_host)} ms | pageable->pinned {f(t_pin)} | H2D+sync {f(t_h2d)} | network_inputs {f(t_prep)} | step {f(t_step)}")
        print("   dtypes:", [(tuple(a.shape), str(a.dtype)) for a in leaves])
