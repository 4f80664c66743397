"""Stage-by-stage GPU diagnosis against the CPU oracle (run on the GPU box):

    PYTHONPATH=. python tests/diag/gpu_diag.py [--quick]

Prints max-abs error (and the reference magnitude) for every encoder stage read back from the
workspace, every decoder head, and the full model, for both precisions; then a few timings.
Never stops at the first failure: one GPU call should tell everything."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from giga_amd import _capi, networks, synth, weights  # noqa: E402
from oracle import giga_oracle as O  # noqa: E402

STAGES = ["P0", "A0", "S0", "Q0", "A1", "S1", "Q1", "A2", "S2", "U0", "A3", "A4", "U1", "A5", "A6", "YZ", "XZ"]
SHAPE = {"P0": (40, 32), "A0": (40, 32), "S0": (40, 32), "Q0": (20, 32), "A1": (20, 64), "S1": (20, 64),
         "Q1": (10, 64), "A2": (10, 128), "S2": (10, 128), "U0": (20, 64), "A3": (20, 64), "A4": (20, 64),
         "U1": (40, 32), "A5": (40, 32), "A6": (40, 32)}


def err(name, got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if got.shape != ref.shape:
        print(f"  {name:28s} SHAPE MISMATCH {tuple(got.shape)} vs {tuple(ref.shape)}")
        return
    d = (got - ref).abs()
    bad = torch.isnan(got).sum().item()
    print(f"  {name:28s} max_err {d.max().item():.3e}  mean_err {d.mean().item():.3e}  "
          f"ref_max {ref.abs().max().item():.3e}  nan {bad}")


def main():
    quick = "--quick" in sys.argv
    dev = torch.device("cuda:0")
    print("device:", torch.cuda.get_device_name(0))
    sd = weights.make_state_dict(7)
    net = networks.get_network("giga")
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    B = 2
    x = torch.from_numpy(synth.tsdf_batch(0, B))
    p = torch.from_numpy(synth.query_points(0, B, 1000, stream=1, half_width=0.6))
    with torch.no_grad():
        feat = O.conv_in_relu(sd, x)
        proj = O.project_planes(feat)
        ref_st = {k: O.unet_stages(sd, proj[k]) for k in O.PLANES}
        ref_planes = {k: ref_st[k]["OUT"] for k in O.PLANES}
        ref_out = O.model_forward(sd, x, p, p_tsdf=p)
        ref_raw = {h: O.decoder_forward(sd, h, p, ref_planes) for h in weights.HEADS}
    L = _capi.lib()
    for prec in ("fp32", "fp16"):
        print(f"===== precision {prec} =====")
        net.set_precision(prec)
        pi = _capi.PRECISION[prec]
        try:
            with torch.no_grad():
                planes = net.encode_inputs(x.to(dev))
                torch.cuda.synchronize()
            offs = (ctypes.c_size_t * 17)()
            L.giga_encoder_workspace_layout(B, pi, offs)
            ws = next(iter(net.encoder._ws.values()))
            es = 2 if pi == 1 else 4
            dt = torch.float16 if pi == 1 else torch.float32
            for i, name in enumerate(STAGES[:15]):
                R, C = SHAPE[name]
                n = 3 * B * R * R * C
                buf = ws[offs[i]:offs[i] + n * es].view(dt).view(3, B, R, R, C)
                if name == "P0":
                    ref = torch.stack([proj[k] for k in O.PLANES])
                else:
                    ref = torch.stack([ref_st[k][name] for k in O.PLANES])
                err("enc." + name, buf.permute(0, 1, 4, 2, 3), ref)
            for i, k in enumerate(O.PLANES):
                err("enc.planes." + k, planes[k], ref_planes[k])
            err("enc.nhwc", planes.nhwc.permute(0, 1, 4, 2, 3), torch.stack([ref_planes[k] for k in O.PLANES]))
        except Exception as e:  # noqa: BLE001
            print("  ENCODER FAILED:", type(e).__name__, e)
        # decoder on ORACLE planes (isolates the decoder)
        try:
            with torch.no_grad():
                dev_planes = {k: ref_planes[k].to(dev) for k in O.PLANES}
                for h in weights.HEADS:
                    getattr(net, h).precision = prec
                    out = getattr(net, h)(p.to(dev), dev_planes)
                    err("dec(oracle planes)." + h, out, ref_raw[h])
                q, r, w = net.decode(p.to(dev), dev_planes)
                err("decode.qual", q, ref_out[0]); err("decode.rot", r, ref_out[1]); err("decode.width", w, ref_out[2])
        except Exception as e:  # noqa: BLE001
            print("  DECODER FAILED:", type(e).__name__, e)
        try:
            with torch.no_grad():
                out = net(x.to(dev), p.to(dev), p_tsdf=p.to(dev))
            for nme, a, b in zip(("qual", "rot", "width", "tsdf"), out, ref_out):
                err("model." + nme, a, b)
        except Exception as e:  # noqa: BLE001
            print("  MODEL FAILED:", type(e).__name__, e)
    if quick:
        return
    # ---- timings ---------------------------------------------------------------------------
    def timeit(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print("===== timings (ms) =====")
    for prec in ("fp32", "fp16"):
        net.set_precision(prec)
        for Bt in (1, 32):
            xt = torch.from_numpy(synth.tsdf_batch(0, Bt)).to(dev)
            blob = net.packed_blob(dev)
            with torch.no_grad():
                t_enc = timeit(lambda: net.encoder.encode_nhwc(xt, blob=blob, precision=prec))
                nhwc, _ = net.encoder.encode_nhwc(xt, blob=blob, precision=prec)
                for N in (2048, 64000):
                    pt = torch.from_numpy(synth.query_points(0, Bt, N)).to(dev)
                    from giga_amd.convonet import decode_heads
                    t3 = timeit(lambda: decode_heads(nhwc, pt, blob, 7, prec, True), n=5)
                    t1 = timeit(lambda: decode_heads(nhwc, pt, blob, 8, prec, False), n=5)
                    gf3 = Bt * N * 154560 / t3 / 1e9
                    print(f"  {prec} B={Bt:3d} N={N:6d}: enc {t_enc:8.3f}  dec3 {t3:8.3f} ({gf3:9.1f} TFLOP/s x1e-3)"
                          f"  dec1 {t1:8.3f}")


if __name__ == "__main__":
    main()
