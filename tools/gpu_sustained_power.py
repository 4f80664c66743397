"""Sustained c4 steps (encoder + lattice decoder, no host synchronisation between steps) for ~3 s per U-Net kernel choice, with
rocm-smi sampled in the background: average socket power and shader clock next to the step / decoder times.
    PYTHONPATH=. python tools/gpu_sustained_power.py [scenes ...]"""
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.convonet import decode_heads
from giga_amd.detection import query_lattice

dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval().set_precision("fp16")
blob = net.packed_blob(dev)
lat = query_lattice(40, dev)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
    except Exception as e:                                   # noqa: BLE001
        return None, None, str(e)
    p = re.search(r"Power[^:]*:\s*([\d.]+)", out)
    c = re.search(r"sclk[^(]*\((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None), (int(c.group(1)) if c else None), out


first = smi()[2]
print("rocm-smi sample:", " | ".join(l.strip() for l in first.splitlines() if "ower" in l or "sclk" in l)[:300], flush=True)
for B in [int(a) for a in sys.argv[1:]] or [32, 128]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    for kernel in ("conv16", "conv32", "conv16", "conv32"):
        net.set_unet_kernel(kernel)
        samples, stop = [], False

        def sampler():
            while not stop:
                p, c, _ = smi()
                samples.append((p, c))
                time.sleep(0.2)
        with torch.no_grad():
            def step():
                nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision="fp16", fold_final=True)
                return decode_heads(nhwc, lat, blob, 7, "fp16", True, folded=True)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            th = threading.Thread(target=sampler); th.start()
            n = max(50, int(3.0 / (B * 12e-6)))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            ev[0].record()
            for i in range(n):
                step(); ev[i + 1].record()
            torch.cuda.synchronize()
            stop = True; th.join()
        ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
        pw = [s[0] for s in samples if s[0]]; ck = [s[1] for s in samples if s[1]]
        print(f"B={B:4d} {kernel}: {n} steps, step median {np.median(ms) * 1e3:8.1f} us, first 10 % {np.median(ms[:n // 10]) * 1e3:8.1f}, last 10 % {np.median(ms[-n // 10:]) * 1e3:8.1f}"
              f"   power {np.mean(pw) if pw else float('nan'):6.0f} W  sclk {np.mean(ck) if ck else float('nan'):6.0f} MHz ({len(samples)} samples)", flush=True)
net.set_unet_kernel("auto")
