"""Which torch ops surround the HIP kernels in one training step (torch.profiler, CPU-side op table)."""
import torch
from torch.profiler import profile, ProfilerActivity
from giga_amd import networks, synth, weights
from giga_amd.training import giga_loss

dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
B, M = 32, 2048
x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
pos_occ = torch.from_numpy(synth.query_points(0, B, M, stream=3)).to(dev)
y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(0, B, M))
opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="count", row_limit=25, max_name_column_width=50))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=12, max_name_column_width=40, max_src_column_width=90))
