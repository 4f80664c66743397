"""Winograd F(2x2, 3x3) layers of the exact-fp32 U-Net (giga_amd/csrc/giga_wino.h; reference encoder/unet.py:14-23,48-114) on the GPU:
every 3x3 layer's output against torch's conv2d + ReLU of the layer's OWN input (read back from the encoder workspace), for the
per-layer launches and for the persistent launch, with every 3x3 layer forced onto the Winograd kernels (GIGA_WINOGRAD is read once
per process, so the all-layers run is a subprocess) and with the default layer set; and the planes against the direct kernels."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from giga_amd import _capi, networks, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["P0", "A0", "S0", "Q0", "A1", "S1", "Q1", "A2", "S2", "U0", "A3", "A4", "U1", "A5", "A6"]
CH = dict(zip(NAMES, (32, 32, 32, 32, 64, 64, 64, 128, 128, 64, 64, 64, 32, 32, 32)))
HW = dict(zip(NAMES, (40, 40, 40, 20, 20, 20, 10, 10, 10, 20, 20, 20, 40, 40, 40)))
TOL = 2e-5         # relative to the layer's largest output: fp32 Winograd sits at a few 1e-6, the direct form at ~1e-6


def layer_errors(sd, Bs, form, first=40):
    """max |layer output - torch conv of the layer's own input| / max |output| for every U-Net stage; also the final planes"""
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(sd); net = net.to(dev).eval().set_precision("fp32")
    net.set_persistent_unet(form)
    x = torch.from_numpy(synth.tsdf_batch(first, Bs))
    with torch.no_grad():
        got = net.encode_inputs(x.to(dev))
    torch.cuda.synchronize()
    path = _capi.lib().giga_encoder_last_path()
    ws = net.encoder._ws.snapshot()[-1]
    off = (ctypes.c_size_t * 17)()
    assert _capi.lib().giga_encoder_workspace_layout(Bs, 0, off) == 0

    def stage(nm):
        n = 3 * Bs * HW[nm] * HW[nm] * CH[nm]
        o = off[NAMES.index(nm)]
        return ws[o:o + 4 * n].view(torch.float32).view(3 * Bs, HW[nm], HW[nm], CH[nm]).permute(0, 3, 1, 2).double().cpu()

    W = lambda k: sd["encoder.unet." + k + ".weight"].double()  # noqa: E731
    Bi = lambda k: sd["encoder.unet." + k + ".bias"].double()  # noqa: E731
    c3 = lambda k, t: F.relu(F.conv2d(t, W(k), Bi(k), padding=1))  # noqa: E731
    up = lambda k, t: F.conv_transpose2d(t, W(k), Bi(k), stride=2)  # noqa: E731
    layers = [("A0", lambda: c3("down_convs.0.conv1", stage("P0"))), ("S0", lambda: c3("down_convs.0.conv2", stage("A0"))),
              ("Q0", lambda: F.max_pool2d(stage("S0"), 2, 2)),
              ("A1", lambda: c3("down_convs.1.conv1", stage("Q0"))), ("S1", lambda: c3("down_convs.1.conv2", stage("A1"))),
              ("Q1", lambda: F.max_pool2d(stage("S1"), 2, 2)),
              ("A2", lambda: c3("down_convs.2.conv1", stage("Q1"))), ("S2", lambda: c3("down_convs.2.conv2", stage("A2"))),
              ("U0", lambda: up("up_convs.0.upconv", stage("S2"))),
              ("A3", lambda: c3("up_convs.0.conv1", torch.cat((stage("U0"), stage("S1")), 1))),
              ("A4", lambda: c3("up_convs.0.conv2", stage("A3"))), ("U1", lambda: up("up_convs.1.upconv", stage("A4"))),
              ("A5", lambda: c3("up_convs.1.conv1", torch.cat((stage("U1"), stage("S0")), 1))),
              ("A6", lambda: c3("up_convs.1.conv2", stage("A5")))]
    errs = {}
    for nm, fn in layers:
        want = fn()
        errs[nm] = float((stage(nm) - want).abs().max() / max(1.0, float(want.abs().max())))
    return errs, path, torch.cat([got[k] for k in ("xz", "xy", "yz")]).cpu()


@pytest.mark.parametrize("form", ["layers", True], ids=["per-layer", "persistent"])
@pytest.mark.parametrize("Bs", [2, 11, 32])
def test_default_winograd_layers(sd7, form, Bs):
    errs, path, _ = layer_errors(sd7, Bs, form)
    assert path & 8, "the fp32 encoder did not take the Winograd kernels"
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, (bad, errs)


def test_planes_match_direct_convolutions(sd7):
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).eval().set_precision("fp32")
    x = torch.from_numpy(synth.tsdf_batch(7, 8)).to(dev)
    with torch.no_grad():
        a = net.set_unet_kernel("auto").encode_inputs(x)
        pa = _capi.lib().giga_encoder_last_path()
        b = net.set_unet_kernel("direct").encode_inputs(x)
        pb = _capi.lib().giga_encoder_last_path()
    assert pa & 8 and not pb & 8
    for k in ("xz", "xy", "yz"):
        d = float((a[k] - b[k]).abs().max())
        assert d < 2e-5 * max(1.0, float(b[k].abs().max())), (k, d)


def test_every_3x3_layer_as_winograd(sd7):
    """GIGA_WINOGRAD=0xFFF: all ten 3x3 layers and the two ConvTranspose GEMM stages (incl. the two-pass 128-channel ones and the 5 x 3 tile blocks of the 20^2 / 10^2 layers)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_wino import layer_errors, TOL\nfrom giga_amd import weights\n"
            "sd = weights.make_state_dict(7)\n"
            "for form in ('layers', True):\n"
            "    for Bs in (2, 11, 32):\n"
            "        errs, path, _ = layer_errors(sd, Bs, form)\n"
            "        print(form, Bs, path, {k: '%%.2e' %% v for k, v in errs.items()})\n"
            "        assert path & 8\n"
            "        assert all(v < TOL for v in errs.values()), errs\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GIGA_WINOGRAD="0xFFF"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-3000:] + r.stderr[-3000:]
