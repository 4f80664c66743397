"""Micro-driver for profiling one workload under rocprofv3:  python tools/gpu_prof.py <what> [iters]
   what: c4dec (f16 decoder, 32 scenes x 64000 lattice points), c2 (fp32 full step), enc16, enc32, b1stages"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from giga_amd import _capi, networks, synth, weights  # noqa: E402
from giga_amd.convonet import decode_heads  # noqa: E402

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
net = networks.get_network("giga"); net.load_state_dict(weights.make_state_dict(7)); net = net.to(dev).eval()
B = 32


def run(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    print(f"{what}: {(time.perf_counter() - t0) / iters * 1e3:.4f} ms/iter")


with torch.no_grad():
    if what == "c4lat":
        from giga_amd.detection import query_lattice
        net.set_precision("fp16"); blob = net.packed_blob(dev)
        x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
        nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision="fp16")
        lat = query_lattice(40, dev)
        run(lambda: decode_heads(nhwc, lat, blob, 7, "fp16", True))
    elif what in ("c4step", "c4step_x3"):                  # the whole c4 step: encoder + lattice resample + fused decoder
        from giga_amd.detection import query_lattice
        prec = "fp16x3" if what.endswith("x3") else "fp16"
        net.set_precision(prec); blob = net.packed_blob(dev)
        x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
        lat = query_lattice(40, dev)

        def c4():
            nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)
            return decode_heads(nhwc, lat, blob, 7, prec, True, folded=True)
        run(c4)
    elif what == "c2_x3":
        net.set_precision("fp16x3")
        x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
        pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
        occ = torch.from_numpy(synth.query_points(0, B, 2048, stream=3)).to(dev)
        run(lambda: net(x, pos, p_tsdf=occ))
    elif what == "c4dec":
        net.set_precision("fp16"); blob = net.packed_blob(dev)
        x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
        nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision="fp16")
        lat = torch.from_numpy(synth.inference_lattice()).to(dev).expand(B, -1, -1).contiguous()
        run(lambda: decode_heads(nhwc, lat, blob, 7, "fp16", True))
    elif what in ("enc16", "enc32"):
        prec = "fp16" if what == "enc16" else "fp32"
        net.set_precision(prec); blob = net.packed_blob(dev)
        x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
        run(lambda: net.encoder.encode_nhwc(x, blob=blob, precision=prec))
    elif what == "c2":
        net.set_precision("fp32")
        x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev)
        pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
        occ = torch.from_numpy(synth.query_points(0, B, 2048, stream=3)).to(dev)
        run(lambda: net(x, pos, p_tsdf=occ))
    elif what == "b1stages":
        L = _capi.lib()
        for prec in ("fp32", "fp16"):
            net.set_precision(prec); blob = net.packed_blob(dev)
            for Bt in (1, 4):
                x = torch.from_numpy(synth.tsdf_batch(0, Bt)).to(dev)
                e0, e1 = L.giga_event_create(), L.giga_event_create()
                ms = ctypes.c_float(); out = []
                for st in range(15):
                    for _ in range(2):
                        net.encoder.encode_nhwc(x, blob=blob, precision=prec, probe=(st, e0, e1))
                    L.giga_event_elapsed_ms(e0, e1, ctypes.byref(ms)); out.append(round(ms.value * 1e3, 1))
                t0 = time.perf_counter()
                for _ in range(10):
                    net.encoder.encode_nhwc(x, blob=blob, precision=prec)
                torch.cuda.synchronize()
                print(prec, "B", Bt, "stage us:", out, "sum", round(sum(out), 1), "wall/iter us", round((time.perf_counter() - t0) / 10 * 1e6, 1))

if what in ("train", "train_bf16", "train_flat", "train_bf16_flat"):      # *_flat: the bench's best leg (flat parameter + giga_amd.optim.FlatAdam)
    from giga_amd.training import giga_loss
    net.train().set_train_precision("bf16" if "bf16" in what else "fp32")
    x = torch.from_numpy(synth.tsdf_batch(0, B)).to(dev); pos = torch.from_numpy(synth.query_points(0, B, 1, stream=2)).to(dev)
    pos_occ = torch.from_numpy(synth.query_points(0, B, 2048, stream=3)).to(dev)
    y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(0, B, 2048))
    if what.endswith("_flat"):
        from giga_amd.optim import FlatAdam
        opt = FlatAdam(net.flatten_parameters(), lr=2e-4)
    else:
        opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)
    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
        loss.backward(); opt.step()
    run(step)
