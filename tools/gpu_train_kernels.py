"""Per-launch kernel durations of ONE training step (rocprofv3 --kernel-trace csv -> table in launch order).
   cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o t --output-format csv -- python tools/gpu_train_kernels.py run [bf16] [detach]
   python tools/gpu_train_kernels.py table /tmp/pt"""
import sys
if sys.argv[1] == "run":
    import torch
    from giga_amd import networks, synth, weights
    from giga_amd.training import giga_loss
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
    net.set_train_precision("bf16" if "bf16" in sys.argv[2:] else "fp32")
    net.detach_tsdf = "detach" in sys.argv[2:]      # occupancy head detached from the planes: no scatter in its backward
    B, M = 32, 2048
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
    pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
    occ = torch.from_numpy(synth.query_points(0, B, M, stream=3)).to(dev)
    y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(0, B, M))
    opt = torch.optim.Adam(net.flatten_parameters(), lr=2e-4, fused=True)
    for i in range(12):
        opt.zero_grad(set_to_none=True); loss, _ = giga_loss(net(x, pos, p_tsdf=occ), y); loss.backward(); opt.step()
    torch.cuda.synchronize()
else:
    import csv, glob
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last step = the kernels after the second-to-last fused-Adam launch
    adam = [i for i, r in enumerate(rows) if "FusedOptimizer" in r["Kernel_Name"] or "fused_adam" in r["Kernel_Name"].lower()]
    lo = adam[-2] + 1 if len(adam) > 1 else 0
    step = rows[lo:adam[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    tot = 0.0
    for r in step:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:8.1f} us  {r['Kernel_Name'][:110]}")
    print(f"kernels {len(step)}, sum of durations {tot:.1f} us, span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
