"""Per-tensor gradient errors of the HIP backward vs torch autograd through the CPU oracle (GPU box)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from giga_amd import networks, synth, weights  # noqa: E402
from giga_amd.training import giga_loss  # noqa: E402
from oracle import giga_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
B, M = 4, 2048
x = torch.from_numpy(synth.tsdf_batch(10, B)); pos = torch.from_numpy(synth.query_points(10, B, 1, stream=2))
pos_occ = torch.from_numpy(synth.query_points(10, B, M, stream=3))
y = tuple(torch.from_numpy(a) for a in synth.train_labels(10, B, M))
sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
rl, _ = O.train_loss(O.train_select(O.model_forward(sdg, x, pos, p_tsdf=pos_occ)), y)
rl.backward()
net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).train()
try:
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    loss, _ = giga_loss(out, tuple(t.to(dev) for t in y))
    print("loss", loss.item(), "ref", rl.item())
    loss.backward()
    torch.cuda.synchronize()
    for name, prm in net.named_parameters():
        ref = sdg[name].grad
        got = prm.grad.detach().cpu()
        sc = ref.abs().max().item()
        err = (got - ref).abs().max().item()
        flag = "" if err <= 2e-3 * sc + 1e-6 else "   <<<<<< BAD"
        print(f"  {name:48s} max|ref| {sc:.3e}  max_err {err:.3e}  rel {err / (sc + 1e-12):.2e}{flag}")
except Exception as e:  # noqa: BLE001
    import traceback; traceback.print_exc()
# timing of a full train step at B=32
try:
    B = 32
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
    pos_occ = torch.from_numpy(synth.query_points(0, B, 2048, stream=3)).to(dev)
    y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(0, B, 2048))
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
        loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"train step B=32 (fwd+bwd+Adam): {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
except Exception as e:  # noqa: BLE001
    import traceback; traceback.print_exc()
