"""SURVEY 8f-3 measurement: is the c5 training step loader-bound?  (run on the GPU box)

    PYTHONPATH=. python tools/gpu_feed_bench.py [n_scenes] [grasps_per_scene]

Writes a synthetic training set in the reference's on-disk layout to a temporary directory, then times
  (a) the training step on RESIDENT inputs (bench.py's c5 leg),
  (b) the same step fed by GraspOccBatches -> TSDFFeed (reader thread + pinned double-buffered H->D staging),
  (c) the reader alone (host batches per second) and (d) the reference-style per-item path (item(i) + stack) for scale."""
import json
import sys
import tempfile
import time

import numpy as np
import torch

from giga_amd import dataset, networks, synth, weights
from giga_amd.feed import TSDFFeed
from giga_amd.training import giga_loss


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    gps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    B, M = 32, 2048
    dev = torch.device("cuda:0")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root, raw = tmp + "/data", tmp + "/raw"
        t0 = time.perf_counter()
        n = synth.write_training_set(root, raw, n_scenes=n_scenes, grasps_per_scene=gps, occ_files=(2, 4), n_occ_points=20000, seed=9)
        out["dataset"] = {"grasps": n, "scenes": n_scenes, "write_s": time.perf_counter() - t0}
        ds = dataset.GraspOccDataset(root, raw, num_point_occ=M)
        net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
        opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)

        def step(x, pos, pos_occ, y):
            opt.zero_grad(set_to_none=True)
            loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
            loss.backward()
            opt.step()

        # (a) resident
        first = next(iter(TSDFFeed(dataset.GraspOccBatches(ds, B, seed=1, drop_last=True), dev)))
        res = dataset.network_inputs(first)
        for _ in range(5):
            step(*res)
        torch.cuda.synchronize()
        K = 60
        t0 = time.perf_counter()
        for _ in range(K):
            step(*res)
        torch.cuda.synchronize()
        out["resident_ms_per_step"] = (time.perf_counter() - t0) / K * 1e3
        # (b) fed: reader thread (shares the GIL with this loop) and persistent reader processes; epoch 1 starts the workers
        # and fills their scene caches, epochs 2-3 are the steady state
        for name, nw in (("thread", 0), ("8_procs_dataloader", 8), ("4_procs_shared_ring", -4), ("8_procs_shared_ring", -8),
                         ("16_procs_shared_ring", -16)):
            src = (dataset.GraspOccRing(ds, B, workers=-nw, seed=2, drop_last=True) if nw < 0 else
                   dataset.GraspOccBatches(ds, B, seed=2, drop_last=True, workers=nw))
            for ep in range(3 if nw else 1):
                k = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for batch in TSDFFeed(src, dev):
                    step(*dataset.network_inputs(batch))
                    k += 1
                torch.cuda.synchronize()
                out[f"fed_ms_per_step_{name}_epoch{ep + 1}"] = (time.perf_counter() - t0) / max(k, 1) * 1e3
            out["steps_per_epoch"] = k
            if nw < 0:
                src.close()
            del src
        # (c) reader alone
        t0 = time.perf_counter(); k = 0
        for _ in dataset.GraspOccBatches(ds, B, seed=3, drop_last=True):
            k += 1
        out["reader_ms_per_batch_warm"] = (time.perf_counter() - t0) / max(k, 1) * 1e3
        # (d) per-item path, the reference's access pattern (single thread, no scene cache reuse across the stack call)
        ds2 = dataset.GraspOccDataset(root, raw, num_point_occ=M, cache_scenes=0)
        t0 = time.perf_counter()
        for i in range(B):
            ds2.item(i)
        out["per_item_ms_per_batch_of_32"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
