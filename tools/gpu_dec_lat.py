"""A/B of the lattice decoders on one GPU: the separable-fc_c kernel (decoder_lat_kernel, GIGA_DEC_LAT=1) against the
previous kernels (GIGA_DEC_LAT=0) for 4 / 8 / 32 / 128 scenes x the 64 000-point lattice, three grasp heads; outputs compared
with each other and (scene 0) with the CPU oracle.  Prints one line per (precision, scenes)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from giga_amd import _capi, networks, synth, weights  # noqa: E402
from giga_amd.convonet import decode_heads  # noqa: E402
from giga_amd.detection import query_lattice  # noqa: E402
from oracle import giga_oracle as O  # noqa: E402

FLOP_GRASP3 = 154_560
dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).eval()
L = _capi.lib()
lat = query_lattice(40, dev)
with torch.no_grad():
    ref = O.model_forward(sd, torch.from_numpy(synth.tsdf_batch(1000, 1)), O.inference_lattice())
ms = ctypes.c_float()
rows = []
sizes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4,8,32,128".split(","))]
for prec in ("fp16", "fp16x3"):
    net.set_precision(prec)
    blob = net.packed_blob(dev)
    for B in sizes:
        x = torch.from_numpy(synth.tsdf_batch(1000, B)).to(dev)
        with torch.no_grad():
            nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)
        res = {}
        for mode in ("0", "1"):
            os.environ["GIGA_DEC_LAT"] = mode
            evs = [(L.giga_event_create(), L.giga_event_create()) for _ in range(12)]
            with torch.no_grad():
                for _ in range(3):
                    out = decode_heads(nhwc, lat, blob, 7, prec, True, folded=True)
                torch.cuda.synchronize()
                for e in evs:
                    out = decode_heads(nhwc, lat, blob, 7, prec, True, probe=e, folded=True)
            torch.cuda.synchronize()
            t = []
            for a, b in evs:
                _capi.check(L.giga_event_elapsed_ms(a, b, ctypes.byref(ms)), "event"); t.append(ms.value)
                L.giga_event_destroy(a); L.giga_event_destroy(b)
            res[mode] = (float(np.median(t)), {k: v.clone() for k, v in out.items()})
        os.environ.pop("GIGA_DEC_LAT", None)
        t0, o0 = res["0"]; t1, o1 = res["1"]
        d_ab = {k: float((o0[k] - o1[k]).abs().max()) for k in o0}
        e_old = {k: float((o0[k][:1].cpu() - r).abs().max()) for k, r in zip(("decoder_qual", "decoder_rot", "decoder_width"), ref)}
        e_new = {k: float((o1[k][:1].cpu() - r).abs().max()) for k, r in zip(("decoder_qual", "decoder_rot", "decoder_width"), ref)}
        fl = B * 64000 * FLOP_GRASP3
        row = {"prec": prec, "scenes": B, "old_ms": round(t0, 4), "new_ms": round(t1, 4), "speedup": round(t0 / t1, 3),
               "old_alg_frac_f16_peak": round(fl / (t0 * 1e-3) / 2.5e15, 3), "new_alg_frac_f16_peak": round(fl / (t1 * 1e-3) / 2.5e15, 3),
               "max_abs_old_vs_new": d_ab, "scene0_err_old": e_old, "scene0_err_new": e_new,
               "finite": bool(all(torch.isfinite(v).all() for v in o1.values()))}
        rows.append(row)
        print(json.dumps(row), flush=True)
