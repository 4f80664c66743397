"""One-off randomized sweep (not collected by pytest): encoder planes at many batch sizes / seeds in all three modes against
the oracle (first, middle, last scene) and against the same scenes run alone.   PYTHONPATH=. python tests/diag/gpu_fuzz_encoder.py"""
import sys
import numpy as np
import torch
from giga_amd import networks, synth, weights
from oracle import giga_oracle as O

dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).eval()
rng = np.random.default_rng(0)
worst = {}
Bs = [1, 2, 3, 7, 8, 9, 15, 16, 24, 31, 32, 33, 39, 40, 47, 48, 56, 63, 64, 65, 96]
for B in Bs:
    seed = int(rng.integers(0, 5000))
    x = torch.from_numpy(synth.tsdf_batch(seed, B)).to(dev)
    # a few adversarial scenes: empty, full, single voxel, checkerboard
    if B >= 4:
        x[0].zero_(); x[1].fill_(1.0); x[2].zero_(); x[2, 0, 0, 0] = 1.0; x[2, 39, 39, 39] = 0.5
        g = torch.arange(40, device=dev)
        x[3] = ((g[:, None, None] + g[None, :, None] + g[None, None, :]) % 2).float()
    ks = sorted({0, 1, 2, 3, B // 2, B - 1} & set(range(B)))
    refs = {k: O.encoder_forward(sd, x[k:k + 1].cpu()) for k in ks}
    for prec, tol in (("fp32", 1e-4), ("fp16x3", 1e-4), ("fp16", 2e-2)):
        net.set_precision(prec)
        for persist in (False, True):
            net.set_persistent_unet(persist)
            with torch.no_grad():
                pl = net.encode_inputs(x)
                for k in ks:
                    one = net.encode_inputs(x[k:k + 1].contiguous())
                    for key in ("xz", "xy", "yz"):
                        scale = max(1.0, float(refs[k][key].abs().max()))
                        e_or = (pl[key][k:k + 1].float().cpu() - refs[k][key]).abs().max().item() / scale
                        e_al = (pl[key][k:k + 1].float() - one[key].float()).abs().max().item() / scale
                        worst[(prec, "oracle")] = max(worst.get((prec, "oracle"), 0), e_or)
                        worst[(prec, "alone")] = max(worst.get((prec, "alone"), 0), e_al)
                        if e_or > tol or e_al > (1e-5 if prec != "fp16" else 1e-2) or not np.isfinite(e_or):
                            print("FAIL", B, prec, persist, k, key, e_or, e_al); sys.exit(1)
    net.set_persistent_unet(False)
    print("B", B, "ok")
print("worst relative errors:", {f"{a}/{b}": f"{v:.2e}" for (a, b), v in worst.items()})
