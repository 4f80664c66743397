"""Which half of the bf16 training step moves the gradients?  PYTHONPATH=. python tests/diag/gpu_bf16_diag.py
Smooth objective (fixed random weights on the head outputs); per-module relative L2 gradient error against fp32 autograd through
the oracle for: fp32 step, bf16 forward only, bf16 data-gradient convolutions only, both."""
import numpy as np
import torch

from giga_amd import _capi, networks, synth, training, weights
from oracle import giga_oracle as O

dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
B, M = 32, 256
x = torch.from_numpy(synth.tsdf_batch(300, B)); pos = torch.from_numpy(synth.query_points(300, B, 1, stream=2))
pos_occ = torch.from_numpy(synth.query_points(300, B, M, stream=3))
rng = torch.Generator().manual_seed(5)
R = [torch.randn(B, 1, generator=rng), torch.randn(B, 1, 4, generator=rng), torch.randn(B, 1, generator=rng), torch.randn(B, M, generator=rng) / 16]
sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
ref_out = O.model_forward(sdg, x, pos, p_tsdf=pos_occ)
sum((o * r).sum() for o, r in zip(ref_out, R)).backward()
ref = {k: v.grad for k, v in sdg.items()}
orig_enc, orig_bwd = _capi.ENC_BF16, _capi.BF16_CONVS
for name, enc, bwd in (("fp32", 0, 0), ("bf16 forward only", orig_enc, 0), ("bf16 dgrad only", 0, orig_bwd), ("bf16 both", orig_enc, orig_bwd)):
    _capi.ENC_BF16, _capi.BF16_CONVS = enc, bwd
    net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).train().set_train_precision("fp32" if name == "fp32" else "bf16")
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    fwd_err = max((o.detach().cpu() - r.detach()).abs().max().item() for o, r in zip(out, ref_out))
    sum((o * r.to(dev)).sum() for o, r in zip(out, R)).backward()
    grp = {}
    for n, p in net.named_parameters():
        e = ((p.grad.cpu() - ref[n]).norm() / (ref[n].norm() + 1e-12)).item()
        key = ".".join(n.split(".")[:4]) if n.startswith("encoder.unet") else n.split(".")[0]
        grp.setdefault(key, []).append(e)
    print(f"{name:20s} forward max abs err {fwd_err:.2e} | " + " ".join(f"{k.replace('encoder.unet.', 'unet.')}:{max(v):.3f}" for k, v in grp.items()))
