"""Micro-timings of the staging primitives.  PYTHONPATH=. python tools/gpu_pin_exp.py"""
import time

import torch

dev = torch.device("cuda:0")
x = torch.rand(32, 40, 40, 40)
pin = torch.empty_like(x).pin_memory()
d = torch.empty_like(x, device=dev)
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()


def t(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)


t("pageable -> pinned memcpy (8.2 MB)", lambda: pin.copy_(x))
t("pageable -> pageable memcpy", lambda: x.clone())
t("pinned -> device, current stream, non_blocking", lambda: d.copy_(pin, non_blocking=True))
t("pageable -> device (.to)", lambda: x.to(dev))


def on_side():
    with torch.cuda.stream(side):
        d.copy_(pin, non_blocking=True)


t("pinned -> device on a side stream", on_side)


def on_side_ev():
    with torch.cuda.stream(side):
        d.copy_(pin, non_blocking=True)
        e = torch.cuda.Event(); e.record(side)
    torch.cuda.current_stream().wait_event(e)
    e.synchronize()


t("pinned -> device on a side stream + event + synchronize", on_side_ev)
big = torch.randn(4096, 4096, device=dev)


def overlapped():
    y = big @ big
    with torch.cuda.stream(side):
        d.copy_(pin, non_blocking=True)
    return y


t("side-stream copy under a 4096^3 matmul", overlapped)
t("4096^3 matmul alone", lambda: big @ big)
import threading


def in_thread():
    th = threading.Thread(target=lambda: pin.copy_(x)); th.start(); th.join()


t("pageable -> pinned memcpy in a thread", in_thread)
