"""Fine timers inside a copy of TSDFFeed's producer (in-memory host batches).  PYTHONPATH=. python tools/gpu_feed_exp3.py"""
import queue
import threading
import time

import numpy as np
import torch

from giga_amd import networks, synth, weights
from giga_amd.training import giga_loss

B, M = 32, 2048
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)
host = []
for i in range(30):
    lab, rot, wid, occ = synth.train_labels(i * B, B, M)
    host.append([synth.tsdf_batch(i * B, B), lab, rot, wid, synth.query_points(i * B, B, 1, stream=2), synth.query_points(i * B, B, M, stream=3), occ])
host = [[torch.from_numpy(a) for a in hb] for hb in host]


def step(b):
    x, lab, rot, wid, pos, pocc, occ = b
    opt.zero_grad(set_to_none=True)
    loss, _ = giga_loss(net(x, pos, p_tsdf=pocc), (lab, rot, wid, occ)); loss.backward(); opt.step()


side = torch.cuda.Stream(dev)
NS = 3
pins = [[torch.empty_like(a).pin_memory() for a in host[0]] for _ in range(NS)]
devs = [[torch.empty_like(a, device=dev) for a in host[0]] for _ in range(NS)]
for variant in ("thread+side", "thread+side", "thread+side, no pinned staging (direct pageable->device)", "main-thread in-line, side stream", "thread, copies on a side stream but NO events"):
    T = {k: [] for k in ("sync", "pincopy", "h2d", "rec", "put")}
    copied, released = [None] * NS, [None] * NS
    out = queue.Queue(maxsize=1)

    def stage(n, hb):
        s = n % NS
        t0 = time.perf_counter()
        if copied[s] is not None and "NO events" not in variant:
            copied[s].synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(side):
            if released[s] is not None and "NO events" not in variant:
                side.wait_event(released[s])
            if "no pinned" in variant:
                t2 = time.perf_counter()
                for d, a in zip(devs[s], hb):
                    d.copy_(a, non_blocking=True)
            else:
                for p_, a in zip(pins[s], hb):
                    p_.copy_(a)
                t2 = time.perf_counter()
                for d, p_ in zip(devs[s], pins[s]):
                    d.copy_(p_, non_blocking=True)
            t3 = time.perf_counter()
            if "NO events" not in variant:
                copied[s] = torch.cuda.Event(); copied[s].record(side)
            t4 = time.perf_counter()
        T["sync"].append(t1 - t0); T["pincopy"].append(t2 - t1); T["h2d"].append(t3 - t2); T["rec"].append(t4 - t3)
        return s

    def producer():
        with torch.cuda.device(dev):
            for n, hb in enumerate(host):
                s = stage(n, hb)
                t0 = time.perf_counter(); out.put(s); T["put"].append(time.perf_counter() - t0)
        out.put(None)

    steps = []
    torch.cuda.synchronize(); t_all = time.perf_counter()
    if variant.startswith("thread"):
        threading.Thread(target=producer, daemon=True).start()
        while True:
            s = out.get()
            if s is None:
                break
            if copied[s] is not None:
                torch.cuda.current_stream().wait_event(copied[s])
            t0 = time.perf_counter(); step(devs[s]); steps.append(time.perf_counter() - t0)
            if "NO events" not in variant:
                released[s] = torch.cuda.Event(); released[s].record(torch.cuda.current_stream())
            else:
                torch.cuda.synchronize()
    else:
        for n, hb in enumerate(host):
            s = stage(n, hb)
            torch.cuda.current_stream().wait_event(copied[s])
            t0 = time.perf_counter(); step(devs[s]); steps.append(time.perf_counter() - t0)
            released[s] = torch.cuda.Event(); released[s].record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    f = lambda v: f"{np.median(v) * 1e3:.2f}/{np.max(v) * 1e3:.1f}" if len(v) else "-"  # noqa: E731
    print(f"{variant}: {(time.perf_counter() - t_all) / len(host) * 1e3:.2f} ms/step | median/max ms: step {f(steps)} sync {f(T['sync'])} pincopy {f(T['pincopy'])} "
          f"h2d {f(T['h2d'])} rec {f(T['rec'])} put {f(T['put'])}", flush=True)
