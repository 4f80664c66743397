"""GPU: the plain-f16 kernels (`precision="fp16"`: f16 operands, fp32 accumulate) held to the oracle evaluated WITH
F16-ROUNDED OPERANDS, i.e. to what the number format itself allows, instead of to the 1e-2 envelope that the format's distance
from fp32 needs (tests/test_f16_error_budget.py).  A kernel defect (a dropped term, a wrong rounding point, a stale operand)
moves results by the f16 floor or more and hides inside 1e-2; it does not hide here.

  * U-Net (encoder/unet.py:225-239) and conv_in + axis means (encoder/voxels.py:57-72,106-107), LAYER BY LAYER: the oracle's
    layer on the GPU's own stage input with f16-rounded weights, fp32 accumulation, output rounded to f16 -- every element
    within ONE f16 ulp (a different summation order can flip the final rounding, nothing else), pooled outputs bit-exact.
  * decoder heads (decoder.py:133-176, layers.py:39-47), two ways.  (A) DATA ON WHICH F16 ROUNDING IS THE IDENTITY: per-scene
    constant integer planes, signed-permutation weight matrices and integer biases keep every feature, hidden activation and
    partial sum a small integer, so plain f16 must return the oracle's integers EXACTLY (the fp32-grade modes to 2e-5: they
    carry the 1e-7 by which bilinear weights miss a sum of one) -- this pins the k-slot
    maps, the fragment order, ReLU, the residual adds and the point -> lane routing of every decoder kernel (generic gather,
    lattice, head-resident, shared-feature; fp32, f16x3, plain f16).  (B) real-valued data in plain f16 against the oracle
    chain with f16-rounded features, weights and hidden activations.  An 11-layer chain of roundings cannot be held to 1e-4:
    two VALID evaluations of that chain (fp32 and fp64 accumulation) already differ by 2e-4 rms / 2.5e-3 max, because a 1e-7
    accumulation-order difference at an f16 rounding boundary flips a hidden activation by a whole f16 ulp (measured here on
    the CPU: a quarter of all points carry at least one flip).  So the kernel is asked to sit as close to both valid
    evaluations as they sit to each other (rms within 1.5x, maximum within 2x), which is half the distance between f16 and fp32
    arithmetic.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from giga_amd import _capi, networks, synth, weights
from oracle import giga_oracle as O

pytestmark = pytest.mark.gpu

h16 = lambda t: t.half().float()  # noqa: E731


def ulp16(v):
    """Spacing of f16 numbers at |v| (normal range 2^-14 .. 65504; 2^-24 below)."""
    a = v.abs().clamp_min(2.0 ** -14)
    return torch.exp2(torch.floor(torch.log2(a)) - 10)


def assert_within_one_ulp(got, want, what):
    """`want` is the fp32 value before the final rounding to f16, `got` the kernel's f16 result."""
    w16 = h16(want)
    diff = (got - w16).abs()
    # one f16 ulp at the value, plus the fp32 accumulation noise of a sum of O(1) terms that cancels (2e-6 of the layer's range:
    # near zero -- ReLU outputs of 1e-5 -- the f16 grid is finer than that noise, 6e-8 in the subnormal range)
    tol = ulp16(torch.maximum(got.abs(), w16.abs())) * 1.001 + 2e-6 * max(1.0, float(want.abs().max()))
    bad = diff > tol
    if bool(bad.any()):
        idx = bad.flatten().nonzero().flatten()[:6]
        rows = [(float(got.flatten()[i]), float(want.flatten()[i]), float(w16.flatten()[i])) for i in idx]
        raise AssertionError((what, "more than one f16 ulp", int(bad.sum()), "of", bad.numel(), "(got, want fp32, want f16):", rows))
    frac = float((diff > 0).float().mean())
    assert frac < 0.02, (what, "fraction of flipped roundings", frac)
    return frac


@pytest.mark.parametrize("kernel", ["conv16", "conv32"])
def test_f16_unet_and_conv_in_layer_by_layer(sd7, kernel):
    """kernel: the conv16 U-Net kernels (GIGA_CONV16_UNET) or the conv32 kernels (the default of the f16-class modes) -- the same
    per-layer contract."""
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).eval().set_precision("fp16").set_unet_kernel(kernel)
    for Bs, first in ((2, 40), (32, 500)):        # the five-x-part conv_in kernels and the one-x-part ones (32 scenes up)
        x = torch.from_numpy(synth.tsdf_batch(first, Bs))
        with torch.no_grad():
            got = net.encode_inputs(x.to(dev))
        torch.cuda.synchronize()
        ws = net.encoder._ws.snapshot()[-1]
        off = (ctypes.c_size_t * 17)()
        assert _capi.lib().giga_encoder_workspace_layout(Bs, _capi.PRECISION["fp16"], off) == 0
        names = ["P0", "A0", "S0", "Q0", "A1", "S1", "Q1", "A2", "S2", "U0", "A3", "A4", "U1", "A5", "A6"]
        ch = dict(zip(names, (32, 32, 32, 32, 64, 64, 64, 128, 128, 64, 64, 64, 32, 32, 32)))
        hw = dict(zip(names, (40, 40, 40, 20, 20, 20, 10, 10, 10, 20, 20, 20, 40, 40, 40)))
        keep = slice(0, 3 * Bs) if Bs <= 2 else torch.tensor([0, 1, Bs - 1, Bs, 2 * Bs - 1, 2 * Bs, 3 * Bs - 1])   # images checked

        def stage(nm):
            n = 3 * Bs * hw[nm] * hw[nm] * ch[nm]
            o = off[names.index(nm)]
            t = ws[o:o + 2 * n].view(torch.float16).view(3 * Bs, hw[nm], hw[nm], ch[nm])
            return t[keep].permute(0, 3, 1, 2).float().cpu()

        W = lambda k: h16(sd7["encoder.unet." + k + ".weight"])  # noqa: E731
        Bi = lambda k: sd7["encoder.unet." + k + ".bias"]  # noqa: E731
        c3 = lambda k, t: F.relu(F.conv2d(t, W(k), Bi(k), padding=1))  # noqa: E731
        up = lambda k, t: F.conv_transpose2d(t, W(k), Bi(k), stride=2)  # noqa: E731
        # conv_in on the f16 MFMA (hi x hi only), ReLU, the three axis means in fp32, planes rounded to f16
        feat = F.relu(F.conv3d(h16(x)[:, None], h16(sd7["encoder.conv_in.weight"]), sd7["encoder.conv_in.bias"], padding=1))
        pl = O.project_planes(feat)
        p0 = torch.cat([pl[k] for k in O.PLANES])[keep]
        assert_within_one_ulp(stage("P0"), p0, ("P0", Bs))
        layers = [("A0", lambda: c3("down_convs.0.conv1", stage("P0"))), ("S0", lambda: c3("down_convs.0.conv2", stage("A0"))),
                  ("A1", lambda: c3("down_convs.1.conv1", stage("Q0"))), ("S1", lambda: c3("down_convs.1.conv2", stage("A1"))),
                  ("A2", lambda: c3("down_convs.2.conv1", stage("Q1"))), ("S2", lambda: c3("down_convs.2.conv2", stage("A2"))),
                  ("U0", lambda: up("up_convs.0.upconv", stage("S2"))),
                  ("A3", lambda: c3("up_convs.0.conv1", torch.cat((stage("U0"), stage("S1")), 1))),
                  ("A4", lambda: c3("up_convs.0.conv2", stage("A3"))), ("U1", lambda: up("up_convs.1.upconv", stage("A4"))),
                  ("A5", lambda: c3("up_convs.1.conv1", torch.cat((stage("U1"), stage("S0")), 1))),
                  ("A6", lambda: c3("up_convs.1.conv2", stage("A5")))]
        for nm, fn in layers:
            assert_within_one_ulp(stage(nm), fn(), (nm, Bs))
        for q, s in (("Q0", "S0"), ("Q1", "S1")):            # the fused 2x2 max-pool: max then round == round then max
            assert torch.equal(stage(q), F.max_pool2d(stage(s), 2, 2)), q
        final = F.conv2d(stage("A6"), W("conv_final"), Bi("conv_final"))
        planes = got.nhwc.reshape(3 * Bs, 40, 40, 32)[keep].permute(0, 3, 1, 2).float().cpu()     # the f16 image the decoders read
        assert_within_one_ulp(planes, final, ("conv_final", Bs))
        nchw = torch.cat([got[k] for k in O.PLANES]).cpu()[keep]        # the reference-layout copy: the same values before rounding
        assert float((nchw - final).abs().max()) <= 2e-6 * max(1.0, float(final.abs().max())) + 1e-6


def _decoder_f16_operands(sd, head, p, c, dt=torch.float32):
    """decoder.py:160-176 with the f16 kernels' rounding points: features, weights and every hidden activation that feeds a
    matrix product are f16; accumulation (in `dt`), the residual stream, biases and fc_p (hi/lo split, exact) are not."""
    r16 = lambda t: t.half().to(dt)  # noqa: E731

    def w(name):
        return sd[f"{head}.{name}.weight"].half().to(dt)

    def b(name):
        return sd[f"{head}.{name}.bias"].to(dt)

    net = F.linear(p.to(dt), sd[f"{head}.fc_p.weight"].to(dt), b("fc_p"))
    c16 = r16(c)
    for i in range(5):
        net = net + F.linear(c16, w(f"fc_c.{i}"), b(f"fc_c.{i}"))
        hid = F.linear(r16(F.relu(net)), w(f"blocks.{i}.fc_0"), b(f"blocks.{i}.fc_0"))
        net = net + F.linear(r16(F.relu(hid)), w(f"blocks.{i}.fc_1"), b(f"blocks.{i}.fc_1"))
    return F.linear(r16(F.relu(net)), w("fc_out"), b("fc_out")).squeeze(-1).float()


def _between_valid_evaluations(got, a, b, what):
    """`a`, `b`: two valid evaluations of the f16-operand chain (fp32 / fp64 accumulation)."""
    got = got.float().cpu()
    ref_rms, ref_max = float((a - b).pow(2).mean().sqrt()), float((a - b).abs().max())
    out = []
    for r in (a, b):
        d = (got - r).abs()
        rms, mx = float(d.pow(2).mean().sqrt()), float(d.max())
        assert rms <= 1.5 * ref_rms + 1e-6 and mx <= 2.0 * ref_max + 1e-5, (what, rms, mx, ref_rms, ref_max)
        out.append((rms, mx))
    return out, (ref_rms, ref_max)


def test_f16_decoder_heads_against_f16_operand_oracle(sd7):
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).eval().set_precision("fp16")
    rng = np.random.default_rng(99)
    B = 3
    planes = {k: h16(torch.from_numpy(rng.standard_normal((B, 32, 40, 40)).astype(np.float32))) for k in O.PLANES}
    planes64 = {k: v.double() for k, v in planes.items()}
    dplanes = {k: v.to(dev) for k, v in planes.items()}
    # (1) generic gather path: random query points (both clamps), every head through LocalDecoder.forward (raw outputs)
    p = torch.from_numpy(synth.query_points(5, B, 4096, stream=1, half_width=0.55))
    c32, c64 = O.sample_features(p, planes), O.sample_features(p.double(), planes64)
    with torch.no_grad():
        for h in weights.HEADS:
            dec = getattr(net, h)
            dec.precision = "fp16"
            got = dec(p.to(dev), dplanes)
            a, b = _decoder_f16_operands(sd7, h, p, c32), _decoder_f16_operands(sd7, h, p, c64, torch.float64)
            print("f16 chain, generic", h, _between_valid_evaluations(got, a, b, ("generic", h)))
    # (2) the inference lattice shared by the three scenes: lattice_resample + the shared-feature multi-head kernel
    #     (decoder_f16_kernel: 18 000 tile-heads); (3) one scene alone: the head-resident kernel (decoder_f16s_kernel)
    from giga_amd.detection import query_lattice
    lat = query_lattice(40, dev)
    with torch.no_grad():
        three = net.decode(lat, dplanes)
        one = net.decode(lat, {k: v[:1].contiguous() for k, v in dplanes.items()})
    pl = O.inference_lattice().expand(B, -1, -1)
    c32, c64 = O.sample_features(pl, planes), O.sample_features(pl.double(), planes64)
    post = {"decoder_qual": torch.sigmoid, "decoder_rot": lambda t: F.normalize(t, dim=2), "decoder_width": lambda t: t}
    for h, g3, g1 in zip(O.GRASP_HEADS, three, one):
        a = post[h](_decoder_f16_operands(sd7, h, pl, c32))
        b = post[h](_decoder_f16_operands(sd7, h, pl, c64, torch.float64))
        print("f16 chain, lattice x3 scenes", h, _between_valid_evaluations(g3, a, b, ("lattice", h)))
        print("f16 chain, lattice, one scene", h, _between_valid_evaluations(g1, a[:1], b[:1], ("lattice-1", h)))


def _integer_state_dict(sd, seed):
    """Heads whose arithmetic is exact in every mode: signed-permutation-like matrices (one +-1 per row), integer biases."""
    g = torch.Generator().manual_seed(seed)
    out = {k: v.clone() for k, v in sd.items()}

    def sparse(rows, cols, per_row=1):
        w = torch.zeros(rows, cols)
        for r in range(rows):
            for c in torch.randperm(cols, generator=g)[:per_row]:
                w[r, c] = 1.0 if torch.rand((), generator=g) < 0.5 else -1.0
        return w

    def ibias(n, lo=-1, hi=2):
        return torch.randint(lo, hi, (n,), generator=g).float()

    for h in weights.HEADS:
        od = out[f"{h}.fc_out.weight"].shape[0]
        out[f"{h}.fc_p.weight"] = torch.zeros(32, 3)
        out[f"{h}.fc_p.bias"] = ibias(32)
        for i in range(5):
            out[f"{h}.fc_c.{i}.weight"] = sparse(32, 96)
            out[f"{h}.fc_c.{i}.bias"] = ibias(32)
            out[f"{h}.blocks.{i}.fc_0.weight"] = sparse(32, 32)
            out[f"{h}.blocks.{i}.fc_0.bias"] = ibias(32)
            out[f"{h}.blocks.{i}.fc_1.weight"] = sparse(32, 32)
            out[f"{h}.blocks.{i}.fc_1.bias"] = ibias(32)
        out[f"{h}.fc_out.weight"] = sparse(od, 32, per_row=3)
        out[f"{h}.fc_out.bias"] = ibias(od)
    return out


@pytest.mark.parametrize("prec", ["fp16", "fp16x3", "fp32"])
def test_decoder_kernels_are_exact_on_integer_data(sd7, prec):
    dev = torch.device("cuda:0")
    sdi = _integer_state_dict(sd7, 11)
    net = networks.get_network("giga"); net.load_state_dict(sdi); net = net.to(dev).eval().set_precision(prec)
    g = torch.Generator().manual_seed(3)
    for B, N in ((8, 77), (3, 2048)):
        vals = torch.tensor([-2.0, -1.0, 1.0, 2.0])
        planes = {k: vals[torch.randint(0, 4, (B, 32, 1, 1), generator=g)].expand(B, 32, 40, 40).contiguous() for k in O.PLANES}
        dplanes = {k: v.to(dev) for k, v in planes.items()}
        p = torch.from_numpy(synth.query_points(9, B, N, stream=2, half_width=0.55))
        with torch.no_grad():
            for h in weights.HEADS:                            # generic gather, single-head launches, raw outputs
                dec = getattr(net, h)
                dec.precision = prec
                want = O.decoder_forward(sdi, h, p, planes)
                # (aten's bilinear weights sum to 1 +- 1e-7, so the oracle itself is 1e-6 off the integers it would return
                #  in exact arithmetic; plain f16 snaps every sampled feature back onto its integer)
                exact = want.round()
                assert float((want - exact).abs().max()) < 1e-5 and float(exact.abs().max()) < 2048
                got = dec(p.to(dev), dplanes).cpu()
                if prec == "fp16":
                    assert torch.equal(got, exact), (prec, h, B, N, float((got - exact).abs().max()))
                else:
                    assert float((got - exact).abs().max()) <= 2e-5, (prec, h, B, N)
    # lattice kernels (a constant plane resamples to itself): one scene = head-resident kernels, several = shared-feature
    from giga_amd.detection import query_lattice
    lat = query_lattice(40, dev)
    for B in (1, 4):
        planes = {k: vals[torch.randint(0, 4, (B, 32, 1, 1), generator=g)].expand(B, 32, 40, 40).contiguous() for k in O.PLANES}
        with torch.no_grad():
            got = net.decode(lat, {k: v.to(dev) for k, v in planes.items()})
            want = O.decode(sdi, O.inference_lattice().expand(B, -1, -1), planes)
        for name, a, b in zip(("qual", "rot", "width"), got, want):
            assert float((a.cpu() - b).abs().max()) <= (5e-6 if prec == "fp16" else 2e-5), (prec, "lattice", B, name)
