"""cProfile of one fed epoch (TSDFFeed over GraspOccBatches with 8 reader processes).  PYTHONPATH=. python tools/gpu_feed_profile.py"""
import cProfile
import pstats
import tempfile
import time

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
    src = dataset.GraspOccBatches(ds, B, seed=2, drop_last=True, workers=8)

    def epoch():
        k = 0
        for batch in TSDFFeed(src, dev):
            x, pos, pocc, y = dataset.network_inputs(batch)
            opt.zero_grad(set_to_none=True)
            loss, _ = giga_loss(net(x, pos, p_tsdf=pocc), y); loss.backward(); opt.step()
            k += 1
        torch.cuda.synchronize()
        return k

    epoch()
    t0 = time.perf_counter(); k = epoch(); print("epoch 2:", (time.perf_counter() - t0) / k * 1e3, "ms/step")
    pr = cProfile.Profile(); pr.enable(); k = epoch(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
