"""Does the U-Net kernel that ran BEFORE the lattice decoder change the decoder's time?  (Round 4: at 64-128 scenes the decoder launch
is 8-9 % slower after the conv32 U-Net than after conv16, although the encoders take the same time.)  Per kernel choice the decoder
launch is timed (a) right behind the encoder, (b) behind the encoder and ~400 us of an idle GPU (one spinning workgroup: clocks and
power recover, caches keep their contents), (c) behind the encoder and a 1-GiB device copy (caches replaced).
    PYTHONPATH=. python tools/gpu_dec_after_enc.py [scenes ...]"""
import sys

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.convonet import decode_heads
from giga_amd.detection import query_lattice

dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval().set_precision("fp16")
blob = net.packed_blob(dev)
lat = query_lattice(40, dev)
big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev); big_b = torch.empty_like(big_a)

for B in [int(a) for a in sys.argv[1:]] or [32, 128]:
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    for kernel in ("conv16", "conv32"):
        net.set_unet_kernel(kernel)
        for what in ("back to back", "idle 400 us", "caches replaced"):
            ts, te = [], []
            with torch.no_grad():
                for it in range(12):
                    e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
                    e0.record()
                    nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision="fp16", fold_final=True)
                    e1.record()
                    if what == "idle 400 us":
                        torch.cuda._sleep(int(400e-6 * 2.4e9))
                    elif what == "caches replaced":
                        big_b.copy_(big_a)
                    e2.record()
                    out = decode_heads(nhwc, lat, blob, 7, "fp16", True, folded=True)
                    e3.record()
                    torch.cuda.synchronize()
                    if it >= 2:
                        te.append(e0.elapsed_time(e1) * 1e3); ts.append(e2.elapsed_time(e3) * 1e3)
            print(f"B={B:4d} {kernel}: {what:16s} encoder {np.median(te):8.1f} us   decoder {np.median(ts):8.1f} us", flush=True)
net.set_unet_kernel("auto")
