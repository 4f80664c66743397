"""Which ATen ops of the training step end in device copies / fills?  torch.profiler table of one bf16 FlatAdam step.
    PYTHONPATH=. python tools/gpu_train_ops.py"""
import torch
from torch.profiler import ProfilerActivity, profile

from giga_amd import networks, synth, weights
from giga_amd.optim import FlatAdam
from giga_amd.training import giga_loss

dev = torch.device("cuda:0")
B, M = 32, 2048
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev)
net.train().set_train_precision("bf16")
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(0, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(0, B, M))
print("label dtypes", [(tuple(t.shape), t.dtype, t.is_contiguous()) for t in y])
opt = FlatAdam(net.flatten_parameters(), lr=2e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    loss.backward(); opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=110))
