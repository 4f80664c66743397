"""In-process A/B of a library knob that is read from the environment PER CALL (e.g. GIGA_C32_PREFETCH): encoder time with the knob
at 0 and at 1, interleaved 0 1 0 1 0 1 per batch size and precision, every call synchronised.
    PYTHONPATH=. python tools/gpu_knob_ab.py GIGA_C32_PREFETCH [B ...]        GIGA_PRECS=fp16,fp16x3"""
import os
import sys

import numpy as np
import torch

from giga_amd import networks, synth, weights

knob = sys.argv[1]
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


for B in [int(a) for a in sys.argv[2:]] or [1, 2, 8, 16, 32, 64, 128]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    for prec in os.environ.get("GIGA_PRECS", "fp16,fp16x3").split(","):
        net.set_precision(prec); blob = net.packed_blob(dev)
        res, planes = {0: [], 1: []}, {}
        with torch.no_grad():
            for rnd in range(3):
                for v in (0, 1):
                    os.environ[knob] = str(v)
                    res[v].append(timed(lambda: net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True), 25))
                    planes[v] = net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)[0].clone()
        a, b = np.median(res[0]), np.median(res[1])
        print(f"B={B:4d} {prec:7s} encoder us: {knob}=0 {a:7.1f}   =1 {b:7.1f}  ({(b / a - 1) * 100:+5.1f} %)   planes identical: {torch.equal(planes[0], planes[1])}"
              f"   rounds {' '.join(f'{r:.1f}' for r in res[0])} | {' '.join(f'{r:.1f}' for r in res[1])}", flush=True)
os.environ.pop(knob, None)
