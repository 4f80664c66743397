"""GPU diagnosis of the f16x3 split decoder (run on the GPU box):

    PYTHONPATH=. python tests/diag/gpu_split_diag.py

Errors of every precision against the CPU oracle (generic gather, lattice path, ragged sizes, foreign planes), then the
decoder launch times of c4 (64 000-point lattice) for B in {1, 8, 32} per precision.  Never stops at the first failure."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from giga_amd import _capi, networks, synth, weights  # noqa: E402
from giga_amd.convonet import decode_heads  # noqa: E402
from giga_amd.detection import query_lattice  # noqa: E402
from oracle import giga_oracle as O  # noqa: E402

FLOP_GRASP3 = 154_560


def main():
    dev = torch.device("cuda:0")
    print("device:", torch.cuda.get_device_name(0))
    sd = weights.make_state_dict(7)
    net = networks.get_network("giga")
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    L = _capi.lib()
    names = ("qual", "rot", "width", "tsdf")
    for B, N in ((2, 1000), (1, 1), (3, 33), (2, 257)):
        x = torch.from_numpy(synth.tsdf_batch(0, B))
        p = torch.from_numpy(synth.query_points(0, B, N, stream=1, half_width=0.6))
        with torch.no_grad():
            ref = O.model_forward(sd, x, p, p_tsdf=p)
            for prec in ("fp32", "fp16", "fp16x3"):
                net.set_precision(prec)
                out = net(x.to(dev), p.to(dev), p_tsdf=p.to(dev))
                torch.cuda.synchronize()
                errs = " ".join(f"{n} {(a.cpu() - b).abs().max().item():.2e}" for n, a, b in zip(names, out, ref))
                print(f"B={B} N={N} {prec:7s} {errs}")
    # raw logits on foreign planes (unfolded images, planes_pack path)
    rng = np.random.default_rng(5)
    rp = {k: torch.from_numpy(rng.standard_normal((2, 32, 40, 40)).astype(np.float32)) for k in ("xz", "xy", "yz")}
    p = torch.from_numpy(synth.query_points(0, 2, 2048, stream=1, half_width=0.6))
    with torch.no_grad():
        for h in weights.HEADS:
            ref = O.decoder_forward(sd, h, p, rp)
            for prec in ("fp32", "fp16", "fp16x3"):
                dec = getattr(net, h)
                dec.precision = prec
                out = dec(p.to(dev), {k: v.to(dev) for k, v in rp.items()})
                print(f"foreign planes {h:14s} {prec:7s} max_err {(out.cpu() - ref).abs().max().item():.2e} "
                      f"(|ref| max {ref.abs().max().item():.2f})")
    # lattice path
    lat = query_lattice(40, dev)
    x1 = torch.from_numpy(synth.tsdf_batch(60, 2))
    with torch.no_grad():
        ref = O.model_forward(sd, x1[1:2], lat.cpu())
        for prec in ("fp32", "fp16", "fp16x3"):
            net.set_precision(prec)
            out = net(x1.to(dev), lat)
            errs = " ".join(f"{n} {(a[1:2].cpu() - b).abs().max().item():.2e}" for n, a, b in zip(names, out, ref))
            print(f"lattice {prec:7s} {errs}")
    # timings: c4 decoder launch (events on the launch stream)
    ms = ctypes.c_float()
    for prec in ("fp16", "fp16x3", "fp32"):
        net.set_precision(prec)
        blob = net.packed_blob(dev)
        for Bc in (1, 8, 32):
            x = torch.from_numpy(synth.tsdf_batch(1000, Bc)).to(dev)
            ev = (L.giga_event_create(), L.giga_event_create())
            ts, enc = [], []
            with torch.no_grad():
                for it in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)
                    e1.record()
                    decode_heads(nhwc, lat, blob, 7, prec, True, probe=ev, folded=True)
                    torch.cuda.synchronize()
                    _capi.check(L.giga_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)), "event")
                    if it >= 2:
                        ts.append(ms.value)
                        enc.append(e0.elapsed_time(e1))
            t = float(np.median(ts))
            tf = Bc * 64000 * FLOP_GRASP3 / (t * 1e-3) / 1e12
            print(f"c4 {prec:7s} B={Bc:3d} decoder {t * 1e3:8.1f} us  {tf:7.1f} TFLOP/s algorithmic  encoder {np.median(enc) * 1e3:7.1f} us")
    # generic (non-lattice) decode at c2(ii) shape
    for prec in ("fp16", "fp16x3", "fp32"):
        net.set_precision(prec)
        blob = net.packed_blob(dev)
        x = torch.from_numpy(synth.tsdf_batch(3000, 32)).to(dev)
        p = torch.from_numpy(synth.query_points(3000, 32, 2048, stream=2)).to(dev)
        ev = (L.giga_event_create(), L.giga_event_create())
        with torch.no_grad():
            nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)
            ts = []
            for it in range(6):
                decode_heads(nhwc, p, blob, 15, prec, True, probe=ev, folded=True)
                torch.cuda.synchronize()
                _capi.check(L.giga_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)), "event")
                ts.append(ms.value)
        t = float(np.median(ts[2:]))
        print(f"generic 32x2048 pts x4 heads {prec:7s} decoder {t * 1e3:8.1f} us  {32 * 2048 * 206016 / (t * 1e-3) / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
