"""A/B of the lattice decoders on one GPU: the separable-fc_c kernel (decoder_lat_kernel, GIGA_DEC_LAT=1) against the
previous kernels (GIGA_DEC_LAT=0) for 4 / 8 / 32 / 128 scenes x the 64 000-point lattice, three grasp heads; outputs compared
with each other and (scene 0) with the CPU oracle.  Prints one line per (precision, scenes)."""
import ctypes
import json
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
dev = torch.device("cuda:0")
sd = weights.make_state_dict(7)
net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).eval()
L = _capi.lib()
lat = query_lattice(40, dev)
with torch.no_grad():
    ref = O.model_forward(sd, torch.from_numpy(synth.tsdf_batch(1000, 1)), O.inference_lattice())
ms = ctypes.c_float()
rows = []
MODES = tuple(os.environ.get("DEC_LAT_MODES", "0,1").split(","))
sizes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4,8,32,128".split(","))]
for prec in ("fp16", "fp16x3"):
    net.set_precision(prec)
    blob = net.packed_blob(dev)
    for B in sizes:
        x = torch.from_numpy(synth.tsdf_batch(1000, B)).to(dev)
        with torch.no_grad():
            nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision=prec, fold_final=True)
        res, times = {}, {m: [] for m in MODES}

        def setmode(mode):
            os.environ["GIGA_DEC_LAT"] = mode.split("n")[0]     # "1n16": separable kernel with 16 waves per workgroup
            os.environ["GIGA_LAT_NW"] = mode.split("n")[1] if "n" in mode else "12"

        with torch.no_grad():
            setmode(MODES[-1])
            for _ in range(max(4, int(30.0 / (0.01 * B + 0.05)))):      # ~30 ms of work: clocks up before anything is timed
                decode_heads(nhwc, lat, blob, 7, prec, True, folded=True)
            torch.cuda.synchronize()
            for rnd in range(4):                                 # modes interleaved: clock / thermal drift hits all of them alike
                for mode in MODES:
                    setmode(mode)
                    evs = [(L.giga_event_create(), L.giga_event_create()) for _ in range(6)]
                    out = decode_heads(nhwc, lat, blob, 7, prec, True, folded=True)
                    for e in evs:
                        out = decode_heads(nhwc, lat, blob, 7, prec, True, probe=e, folded=True)
                    torch.cuda.synchronize()
                    for a, b in evs:
                        _capi.check(L.giga_event_elapsed_ms(a, b, ctypes.byref(ms)), "event"); times[mode].append(ms.value)
                        L.giga_event_destroy(a); L.giga_event_destroy(b)
                    res[mode] = (0.0, {k: v.clone() for k, v in out.items()})
        for mode in MODES:
            res[mode] = (float(np.median(times[mode])), res[mode][1])
        os.environ.pop("GIGA_DEC_LAT", None); os.environ.pop("GIGA_LAT_NW", None)
        t0, o0 = res["0"]
        fl = B * 64000 * FLOP_GRASP3
        names = ("decoder_qual", "decoder_rot", "decoder_width")
        row = {"prec": prec, "scenes": B}
        for mode in MODES:
            t, o = res[mode]
            row[f"ms_{mode}"] = round(t, 4)
            row[f"alg_frac_{mode}"] = round(fl / (t * 1e-3) / 2.5e15, 3)
            row[f"err_{mode}"] = [float("%.2e" % float((o[k][:1].cpu() - r).abs().max())) for k, r in zip(names, ref)]
            row[f"vs0_{mode}"] = [float("%.2e" % float((o0[k] - o[k]).abs().max())) for k in names]
            row[f"finite_{mode}"] = bool(all(torch.isfinite(v).all() for v in o.values()))
        rows.append(row)
        print(json.dumps(row), flush=True)
