"""GPU: the conv32 U-Net kernels (csrc/giga_conv32.h; the default of 'fp16' / 'fp16x3' since round 4; `net.set_unet_kernel`, C ABI flags
GIGA_CONV32_UNET / GIGA_CONV16_UNET) held to the same contracts as the conv16 kernels (encoder/unet.py:225-239):
  * the kernel-choice flags reach the launcher (giga_encoder_last_path), in both launch forms, with and without fused pairs;
  * fp16x3 planes and head outputs against the oracle at the fp32 tolerance (1e-4); plain fp16 planes inside the f16 envelope
    (the per-layer one-ulp test is tests/test_gpu_f16_exact.py::test_f16_unet_and_conv_in_layer_by_layer[conv32]);
  * the persistent launch == per-layer launches, BIT FOR BIT (the summation order of an output is fixed: bias, taps x k-chunks),
    at batch sizes that put one image, none, or several images unevenly on a group, and at 128 scenes, where a member's band is
    walked in several sub-bands -- also against the same scenes run as four batches of 32;
  * conv32 against conv16 on the same input: fp16x3 to 2e-5 of the planes' range (both are fp32-grade evaluations).
The data movement itself (staging, zero padding, tiles, fragment order, slice groups, channel parts) is checked without a GPU by
tests/test_conv32_emulation.py."""
import os

import pytest
import torch

from giga_amd import networks, synth
from oracle import giga_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net32(sd7):
    n = networks.get_network("giga")
    n.load_state_dict(sd7)
    return n.to(torch.device("cuda:0")).eval().set_unet_kernel("conv32")


def _planes(net, x):
    with torch.no_grad():
        fea = net.encode_inputs(x)
    return torch.stack([fea[k] for k in O.PLANES]).clone()


def test_kernel_choice_flags_select_what_runs(net32):
    """`set_unet_kernel` / GIGA_CONV32_UNET / GIGA_CONV16_UNET must reach the launcher (a flag that is masked and dropped on the way
    leaves every comparison in this file comparing a kernel with itself): giga_encoder_last_path() says what the call ran."""
    from giga_amd import _capi
    L = _capi.lib()
    dev = torch.device("cuda:0")
    try:
        for prec in ("fp16", "fp16x3"):
            net32.set_precision(prec)
            for B, fused in ((1, True), (11, True), (32, False)):
                x = torch.from_numpy(synth.tsdf_batch(3, B)).to(dev)
                env = os.environ.get("GIGA_CONV32")              # a process-wide override of "auto" (the flag of a call still wins)
                auto32 = (B <= 16) if env is None else (int(env) != 0)
                for kernel, want in (("conv16", 0), ("conv32", _capi.PATH_CONV32), ("auto", _capi.PATH_CONV32 if auto32 else 0)):
                    net32.set_unet_kernel(kernel)
                    for form, pbit in ((False, _capi.PATH_PERSISTENT), ("layers", 0)):
                        net32.set_persistent_unet(form)
                        _planes(net32, x)
                        got = L.giga_encoder_last_path()
                        # (f16x3 launches the fused-pairs instantiation too; none of its pairs fits two weight sets side by side)
                        exp = want | pbit | (_capi.PATH_FUSED_PAIRS if want and pbit and fused else 0)
                        assert got == exp, (prec, B, kernel, form, got, exp)
        net32.set_precision("fp32")                       # fp32 has conv16 only, whatever is asked for
        net32.set_unet_kernel("conv32").set_persistent_unet(False)
        _planes(net32, torch.from_numpy(synth.tsdf_batch(3, 32)).to(dev))
        assert L.giga_encoder_last_path() == _capi.PATH_PERSISTENT | _capi.PATH_WINOGRAD     # (fp32: 3x3 layers as Winograd, giga_wino.h)
    finally:
        net32.set_unet_kernel("conv32").set_persistent_unet(False).set_precision("fp32")


def test_conv32_against_oracle_and_conv16(net32, sd7):
    dev = torch.device("cuda:0")
    x = torch.from_numpy(synth.tsdf_batch(70, 3))
    p = torch.from_numpy(synth.query_points(70, 3, 512, stream=4, half_width=0.55))
    with torch.no_grad():
        ref_planes = O.encoder_forward(sd7, x)
        ref = O.model_forward(sd7, x, p, p_tsdf=p)
    want = torch.stack([ref_planes[k] for k in O.PLANES])
    scale = float(want.abs().max())
    try:
        for prec, tol_p, tol_o in (("fp16x3", 1e-4, 1e-4), ("fp16", 3e-2, 2e-2)):
            net32.set_precision(prec)
            got = _planes(net32, x.to(dev)).cpu()
            assert float((got - want).abs().max()) <= tol_p * max(1.0, scale), (prec, float((got - want).abs().max()), scale)
            with torch.no_grad():
                out = net32(x.to(dev), p.to(dev), p_tsdf=p.to(dev))
            for name, a, b in zip(("qual", "rot", "width", "tsdf"), out, ref):
                assert float((a.cpu() - b).abs().max()) < tol_o * (2.0 if name in ("rot", "width") else 1.0), (prec, name)
            net32.set_unet_kernel("conv16")
            other = _planes(net32, x.to(dev)).cpu()
            net32.set_unet_kernel("conv32")
            if prec == "fp16x3":
                assert float((got - other).abs().max()) <= 2e-5 * max(1.0, scale)
    finally:
        net32.set_unet_kernel("conv32")
        net32.set_precision("fp32")


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8, 11, 32])
def test_conv32_persistent_launch_is_bit_identical_to_per_layer_launches(net32, B):
    dev = torch.device("cuda:0")
    x = torch.from_numpy(synth.tsdf_batch(520, B)).to(dev)
    try:
        for prec in ("fp16x3", "fp16"):
            net32.set_precision(prec)
            got = {}
            for flag in ("layers", False):
                net32.set_persistent_unet(flag)
                for _ in range(2):
                    got[flag] = _planes(net32, x)
            assert torch.equal(got["layers"], got[False]), prec
            one = _planes(net32, x[B - 1:B].contiguous())          # the last scene alone: another work distribution (conv_in's summation
            if prec == "fp16x3":                                   # order depends on the batch size: compare at rounding level)
                assert float((one[:, 0] - got[False][:, B - 1]).abs().max()) <= 2e-5 * max(1.0, float(one.abs().max()))
    finally:
        net32.set_persistent_unet(False)
        net32.set_precision("fp32")


def test_conv32_large_batch_sub_bands(net32):
    """128 scenes: 12 images per group, a member's band (61 rows at 40 x 40) is walked in several sub-bands; the planes equal the
    same scenes encoded as four batches of 32 bit for bit in the f16-class modes."""
    dev = torch.device("cuda:0")
    x = torch.from_numpy(synth.tsdf_batch(1000, 128)).to(dev)
    try:
        for prec in ("fp16", "fp16x3"):
            net32.set_precision(prec)
            full = _planes(net32, x)
            for c0 in range(0, 128, 32):
                part = _planes(net32, x[c0:c0 + 32].contiguous())
                assert torch.equal(full[:, c0:c0 + 32], part), (prec, c0)
    finally:
        net32.set_precision("fp32")
