"""CPU tests: the C-ABI library loads and exports every symbol include/giga_hip.h declares, the
host-side packer and module tree behave (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from giga_amd import _capi, networks, weights
from giga_amd.convonet import (ConvolutionalOccupancyNetwork, ConvolutionalOccupancyNetworkGeometry,
                               LocalDecoder, LocalVoxelEncoder)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "giga_hip.h")).read()
    declared = set(re.findall(r"\b(giga_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libgiga_hip.so does not export {name}"
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    assert lib.giga_abi_version() == 3
    assert lib.giga_strerror(0) == b"ok" and lib.giga_strerror(-4) == b"workspace too small"


def test_python_constants_equal_the_header_defines():
    """giga_amd._capi repeats the flag values of include/giga_hip.h; pin them to the header (a drifted flag bit would select
    another launch form or arithmetic mode silently), and check that the library masks every flag before validating."""
    hdr = open(os.path.join(ROOT, "include", "giga_hip.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(GIGA_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)\s*$", hdr, re.M)}
    for name, value in (("GIGA_FOLD_FINAL", _capi.FOLD_FINAL), ("GIGA_PERSIST_UNET", _capi.PERSIST_UNET),
                        ("GIGA_LAYERWISE_UNET", _capi.LAYERWISE_UNET), ("GIGA_MAX_SCENES", _capi.MAX_SCENES),
                        ("GIGA_CONV32_UNET", _capi.CONV32_UNET), ("GIGA_CONV16_UNET", _capi.CONV16_UNET)):
        assert defines.get(name) == value, (name, defines.get(name), value)
    flags = _capi.FOLD_FINAL | _capi.PERSIST_UNET | _capi.LAYERWISE_UNET | _capi.CONV32_UNET | _capi.CONV16_UNET
    assert flags & 3 == 0 and bin(flags).count("1") == 5          # distinct bits above the precision values 0..3
    lib = _capi.lib()
    for f in (_capi.PERSIST_UNET, _capi.LAYERWISE_UNET, _capi.CONV32_UNET, _capi.CONV16_UNET, flags):
        assert lib.giga_encoder_workspace_bytes(4, 1 | f) == lib.giga_encoder_workspace_bytes(4, 1)
        assert lib.giga_encoder_forward(None, None, None, None, 0, f, None, 0, None) == 0          # empty batch
    # the launch form of the U-Net is a per-call choice of the module: False (default), True (forced persistent), "layers"
    net = networks.get_network("giga")
    for mode in (True, "layers", False):
        assert net.set_persistent_unet(mode) is net and net.encoder.persistent_unet == mode
    # ... and so are the U-Net kernels of the f16-class modes: "auto" (the library's default), "conv32", "conv16"
    for kernel in ("conv32", "conv16", "auto"):
        assert net.set_unet_kernel(kernel) is net and net.encoder.unet_kernel == kernel
    with pytest.raises(ValueError):
        net.set_unet_kernel("conv64")


def test_forget_device_state_is_callable_without_a_gpu():
    """giga_forget_device_state (for hosts that call hipDeviceReset) only clears host-side tables: callable anywhere, any number of times."""
    lib = _capi.lib()
    lib.giga_forget_device_state()
    lib.giga_forget_device_state()
    assert lib.giga_encoder_last_path() in range(8)


def test_param_counts_and_sizes():
    lib = _capi.lib()
    assert lib.giga_param_count(15) == 581863
    assert lib.giga_param_count(7) == 581863 - 26241   # giga_aff: no occupancy head
    assert lib.giga_param_count(8) == 476800 + 26241   # giga_geo
    assert lib.giga_param_count(0) == 476800
    assert lib.giga_packed_bytes() > 0
    assert lib.giga_encoder_workspace_bytes(0, 0) == 0
    assert lib.giga_encoder_workspace_bytes(32, 1) < lib.giga_encoder_workspace_bytes(32, 0)


def test_argument_validation_without_gpu():
    lib = _capi.lib()
    assert lib.giga_encoder_forward(None, None, None, None, 1, 0, None, 0, None) == -1
    assert lib.giga_encoder_forward(None, None, None, None, 0, 0, None, 0, None) == 0     # empty batch
    assert lib.giga_decoder_forward(None, None, None, 7, None, None, None, None, 0, 5, 0, 1, None) == 0
    assert lib.giga_decoder_forward(None, None, None, 7, None, None, None, None, 1, 5, 4, 1, None) == -5
    assert lib.giga_decoder_forward(None, None, None, 7, None, None, None, None, 1, 5, 3, 1, None) == -1      # bf16 decoder: null pointers
    assert lib.giga_pack_weights(None, 0, 15, None, 0) == -1
    # more than GIGA_MAX_SCENES scenes per call: refused before anything is enqueued (fake host addresses, never dereferenced)
    fake = torch.zeros(4)
    assert _capi.MAX_SCENES == 3072 and lib.giga_strerror(-7).startswith(b"more than GIGA_MAX_SCENES")
    assert lib.giga_encoder_forward(_capi.ptr(fake), _capi.ptr(fake), _capi.ptr(fake), None, _capi.MAX_SCENES + 1, 0,
                                    _capi.ptr(fake), 1 << 60, None) == -7
    flat = torch.zeros(10)
    with pytest.raises(_capi.GigaHipError):
        _capi.pack_weights(flat, 15)
    # the flag bits of `precision` / `head_present` are masked before validation; unknown precisions still fail
    assert lib.giga_decoder_forward(None, None, None, 7, None, None, None, None, 0, 5, _capi.FOLD_FINAL, 1, None) == 0
    assert lib.giga_decoder_forward(None, None, None, 7, None, None, None, None, 1, 5, 3 | _capi.FOLD_FINAL, 1, None) == -5
    assert lib.giga_encoder_workspace_bytes(4, _capi.FOLD_FINAL) == lib.giga_encoder_workspace_bytes(4, 0)
    assert lib.giga_backward_workspace_bytes(2, 1, 64, 15 | _capi.DETACH_OCC) == lib.giga_backward_workspace_bytes(2, 1, 64, 15)
    # grasp post-processing: null pointers, bad sizes
    import ctypes
    prm = _capi.GraspParams(1.0, 0.033, 0.233, 0.5, 0.5, 0.9, 2, 2, 7, 4, 0)
    assert lib.giga_grasp_select(None, None, None, None, 1, 40, ctypes.byref(prm), None, None, 16, None, None, None, None,
                                 None, 0, None) == -6
    assert lib.giga_grasp_workspace_bytes(0, 40) == 0 and lib.giga_grasp_workspace_bytes(2, 40) == 2 * 2 * 64000 * 4
    assert lib.giga_packed_bytes() > 4_000_000          # both precisions, plain and folded head images
    # giga_backward validates before it enqueues anything: fake (host) addresses are never dereferenced here
    host = (ctypes.c_float * 16)()
    fake = ctypes.addressof(host)
    null4 = (ctypes.c_void_p * 4)()                     # outs / douts with every head pointer NULL
    n15 = lib.giga_param_count(15)
    bw = lambda **k: lib.giga_backward(fake, fake, fake, fake, fake, fake, k.get("p_tsdf", fake), k.get("outs", null4),
                                       k.get("douts", null4), fake, k.get("n", n15), k.get("heads", 15), k.get("B", 2),
                                       k.get("N", 1), k.get("M", 8), fake, k.get("wsb", 0), None)
    assert bw(B=0) == 0                                 # empty batch
    assert bw(N=-1) == -1 and bw(M=-3) == -1
    assert bw(n=n15 - 1) == -2 and bw(heads=7) == -2    # parameter count must match the head set
    assert bw() == -6                                   # a head that runs needs its out and dout pointers
    some = (ctypes.c_void_p * 4)(fake, fake, fake, None)
    assert bw(outs=some, douts=some) == -6              # ... the occupancy head too (M > 0, p_tsdf given)
    assert bw(outs=some, douts=some, p_tsdf=None) == -4 # without occupancy queries that head does not run; next check
    assert lib.giga_encoder_workspace_layout(2, 7, (ctypes.c_size_t * 17)()) == -5
    assert lib.giga_decoder_forward_lattice(fake, fake, fake, 1, None, None, None, None, 1, 4, 0, 1, fake, 1 << 30, None,
                                            None, None) == -6


def test_pack_is_deterministic_and_sensitive(sd7):
    flat = torch.cat([v.reshape(-1) for v in sd7.values()])
    a = _capi.pack_weights(flat, 15)
    b = _capi.pack_weights(flat.clone(), 15)
    assert torch.equal(a, b)
    flat2 = flat.clone(); flat2[12345] += 1.0
    assert not torch.equal(a, _capi.pack_weights(flat2, 15))


def test_pack_map_reproduces_fp32_words(sd7):
    """Training path: blob fp32 words are pure gathers; applying giga_pack_map in numpy must equal the
    host-packed blob on every fp32 word, for the full and the reduced head sets."""
    for head_present, sd in ((15, sd7), (7, weights.make_state_dict(3, with_tsdf=False))):
        flat = torch.cat([v.reshape(-1) for v in sd.values()])
        blob = _capi.pack_weights(flat, head_present).numpy().view(np.float32)
        m = _capi.pack_map(head_present).numpy()
        assert m.shape == blob.shape and m.max() < flat.numel()
        sel = m >= 0
        np.testing.assert_array_equal(blob[sel], flat.numpy()[m[sel]])
        assert np.all(blob[m == -1] == 0)
        assert sel.sum() > 500000 and (m == -2).sum() > 0
        # every parameter of every present head / the encoder is referenced by the fp32 image
        assert np.unique(m[sel]).size == flat.numel()


def test_conv_fragments_round_trip(sd7):
    """Invert the documented 16x16-MFMA fragment layout (giga_pack.cpp) and recover the weights."""
    flat = torch.cat([v.reshape(-1) for v in sd7.values()])
    blob = _capi.pack_weights(flat, 15).numpy()
    at = 14 * 64 * 4 + 256
    W = sd7["encoder.unet.down_convs.0.conv1.weight"].numpy()         # (32,32,3,3), layer 0
    # layer 0: nb16 = 2, taps = 9; f16: kg = 1 (32 channels / fragment), f32: kg = 2 (16 channels / fragment)
    f16 = blob[at:at + 18 * 1024].view(np.float16).reshape(2, 9, 1, 64, 8)      # [nb][tap][kg][lane][e]
    f32 = blob[at + 18 * 1024:at + 54 * 1024].view(np.float32).reshape(2, 9, 2, 64, 4)
    for nb in range(2):
        for tap in (0, 4, 8):
            for lane in (0, 17, 40, 63):
                j, g = lane & 15, lane >> 4
                co = nb * 16 + j
                for kg in range(2):
                    np.testing.assert_array_equal(
                        f32[nb, tap, kg, lane], W[co, kg * 16 + 4 * g:kg * 16 + 4 * g + 4, tap // 3, tap % 3])
                np.testing.assert_array_equal(
                    f16[nb, tap, 0, lane], W[co, 8 * g:8 * g + 8, tap // 3, tap % 3].astype(np.float16))
    # conv_in B operands: [half][s][lane] = W[16*half + (lane&15)][tap 4s + (lane>>4)], tap 27 -> 0
    ci = blob[:14 * 64 * 4].view(np.float32).reshape(2, 7, 64)
    Wi = sd7["encoder.conv_in.weight"].numpy().reshape(32, 27)
    assert ci[0, 3, 5] == Wi[5, 12] and ci[1, 3, 37] == Wi[16 + 5, 14] and ci[1, 6, 63] == 0.0
    assert ci[0, 6, 40] == Wi[8, 26]
    # ConvTranspose layer (index 6): W[ci][co][dy][dx], sub-output d = dy*2+dx is the fragment "sub"
    off = at
    convs = [(9, 32, 32), (9, 32, 32), (9, 32, 64), (9, 64, 64), (9, 64, 128), (9, 128, 128)]
    for taps, cin, cout in convs:
        nblk = cout // 16
        off += nblk * taps * (cin // 32) * 1024 + nblk * taps * (cin // 16) * 1024 + (cout * 4 + 255) // 256 * 256
    Wu = sd7["encoder.unet.up_convs.0.upconv.weight"].numpy()         # (128, 64, 2, 2)
    n16 = 4 * 4 * 1 * 4                                               # subs * nb16 * taps * (128/32)
    f32u = blob[off + n16 * 1024:off + n16 * 1024 + 4 * 4 * 8 * 1024].view(np.float32).reshape(4, 4, 1, 8, 64, 4)
    for sub, nb, kg, lane in ((0, 0, 0, 3), (3, 2, 5, 44), (1, 3, 7, 63)):
        j, g = lane & 15, lane >> 4
        np.testing.assert_array_equal(f32u[sub, nb, 0, kg, lane],
                                      Wu[kg * 16 + 4 * g:kg * 16 + 4 * g + 4, nb * 16 + j, sub // 2, sub % 2])


def test_module_tree_matches_reference_state_dict():
    for name, nheads in (("giga", 4), ("giga_aff", 3), ("giga_detach", 4)):
        net = networks.get_network(name)
        assert isinstance(net, ConvolutionalOccupancyNetwork)
        keys = list(net.state_dict().keys())
        ref = list(weights.giga_param_shapes(with_tsdf=nheads == 4).keys())
        assert keys == ref
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == weights.giga_param_shapes()[k]
    geo = networks.get_network("giga_geo")
    assert isinstance(geo, ConvolutionalOccupancyNetworkGeometry)
    assert list(geo.state_dict().keys()) == list(weights.giga_param_shapes(heads=("decoder_tsdf",)).keys())
    with pytest.raises(NotImplementedError):
        networks.get_network("vgn")


def test_reference_init_conventions():
    net = networks.get_network("giga")
    assert float(net.decoder_qual.blocks[0].fc_1.weight.abs().max()) == 0.0     # layers.py:37
    assert float(net.encoder.unet.conv_final.bias.abs().max()) == 0.0           # unet.py:216
    assert isinstance(net.encoder, LocalVoxelEncoder) and isinstance(net.decoder_rot, LocalDecoder)
    assert net.decoder_rot.fc_out.weight.shape == (4, 32)


def test_cpu_tensors_fail_loudly():
    net = networks.get_network("giga").eval()
    with torch.no_grad():
        with pytest.raises(_capi.GigaHipError):
            net(torch.zeros(1, 40, 40, 40), torch.zeros(1, 4, 3))
        with pytest.raises(_capi.GigaHipError):
            net.encoder(torch.zeros(1, 40, 40, 40))
    with pytest.raises(_capi.GigaHipError):           # the differentiable path is HIP-only as well
        net(torch.zeros(1, 40, 40, 40), torch.zeros(1, 4, 3))
    with pytest.raises(NotImplementedError):          # piecewise entry points are inference-only
        net.encoder(torch.zeros(1, 40, 40, 40))


def test_unsupported_configs_are_rejected():
    with pytest.raises(NotImplementedError):
        LocalDecoder(c_dim=128, hidden_size=256)
    with pytest.raises(NotImplementedError):
        LocalVoxelEncoder(c_dim=32, unet=False)


def test_load_network_round_trip(tmp_path, sd7):
    path = tmp_path / "vgn_giga_7.pt"
    torch.save(sd7, path)
    net = networks.load_network(path, "cpu")              # model name parsed from the stem (networks.py:28-31)
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd7[k])


def test_feed_and_generation_have_no_cpu_path():
    from giga_amd._capi import GigaHipError
    from giga_amd.feed import TSDFFeed
    with pytest.raises(GigaHipError):
        TSDFFeed([], device="cpu")


def test_ordered_param_cache_tracks_replaced_parameters(sd7):
    net = networks.get_network("giga")
    assert [id(p) for p in net._ordered_params()] == [id(p) for p in net.parameters()]
    assert net._ordered_params() is net._ordered_params()              # cached
    old = net._ordered_params()
    net.load_state_dict(sd7, assign=True)                              # replaces every Parameter object
    new = net._ordered_params()
    assert [id(p) for p in new] == [id(p) for p in net.parameters()] and new is not old
    assert torch.equal(new[0], sd7["decoder_qual.fc_c.0.weight"])
    net.double()                                                       # _apply drops the cache as well
    assert net._ordered_params() is not new and net._ordered_params()[0].dtype == torch.float64


def test_packed_blob_stamp_and_check():
    """Both packers stamp their blob (magic, ABI version, size) and giga_packed_check refuses a blob of another layout: a stale
    blob from a previous build would otherwise be read at the wrong offsets, silently (the compute entry points take no size)."""
    import torch
    from giga_amd import weights
    lib = _capi.lib()
    flat = torch.cat([v.reshape(-1) for v in weights.make_state_dict(3).values()])
    blob, bblob = _capi.pack_weights(flat, 15), _capi.pack_bwd_weights(flat, 15)
    assert lib.giga_packed_check(_capi.ptr(blob), blob.numel(), 0) == 0
    assert lib.giga_packed_check(_capi.ptr(bblob), bblob.numel(), 1) == 0
    assert lib.giga_packed_check(_capi.ptr(bblob), bblob.numel(), 0) == -8          # a backward blob is not a forward blob
    assert lib.giga_packed_check(_capi.ptr(blob), blob.numel() - 256, 0) == -8      # truncated (e.g. a blob of the previous layout)
    stale = blob.clone(); stale[-256 + 8] = 1                                       # ABI version 1 in the stamp
    assert lib.giga_packed_check(_capi.ptr(stale), stale.numel(), 0) == -8
    assert lib.giga_packed_check(None, 0, 0) == -1 and lib.giga_strerror(-8).startswith(b"packed blob was not produced")


def test_precision_tables_cover_every_named_mode():
    """Every name of `net.set_precision` maps to an encoder precision, a generic-decoder precision, a lattice-decoder precision
    (with its flags) and a plane element type; the mixed mode 'fp16x3+fp16' is the f16x3 encoder (fp32 planes) under the plain-f16
    lattice decoder with GIGA_PLANES_FP32 (include/giga_hip.h) and the f16x3 decoder for other query sets."""
    import torch
    from giga_amd import _capi
    for name, v in _capi.PRECISION.items():
        for table in (_capi.ENCODER_PRECISION, _capi.DECODER_PRECISION, _capi.LATTICE_PRECISION, _capi.PLANE_DTYPE):
            assert v in table, (name, table)
        assert _capi.PLANE_DTYPE[v] == _capi.PLANE_DTYPE[_capi.ENCODER_PRECISION[v]], name     # planes are the ENCODER's
    m = _capi.PRECISION["fp16x3+fp16"]
    assert _capi.ENCODER_PRECISION[m] == 2 and _capi.DECODER_PRECISION[m] == 2 and _capi.PLANE_DTYPE[m] == torch.float32
    assert _capi.LATTICE_PRECISION[m] == (1 | _capi.PLANES_FP32) and _capi.PLANES_FP32 == 1024
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "giga_hip.h")).read()
    assert "#define GIGA_PLANES_FP32 1024" in hdr


def test_every_header_is_a_build_dependency():
    """giga_amd/csrc/Makefile: editing ANY header of csrc/ (or include/giga_hip.h) makes `make` want to recompile.  Round 5's
    hand-kept HDRS list had lost giga_side.h: build.py noticed the edit, called make, and make rebuilt nothing.
    `make -n -W file` treats `file` as just modified without touching it."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "giga_amd", "csrc")
    hdrs = sorted(glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "include", "giga_hip.h")]
    assert any(h.endswith("giga_side.h") for h in hdrs) and len(hdrs) >= 11
    for h in hdrs:
        rel = os.path.relpath(h, csrc)
        r = subprocess.run(["make", "-n", "-W", rel], cwd=csrc, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.count("hipcc") >= 2 and "-shared" in r.stdout, f"{rel}: a change would not rebuild the library"


def test_convin_mask_flag_is_refused_for_precisions_that_store_no_mask():
    """GIGA_CONVIN_MASK with precision 1 / 2 (f16-class encoders store no ReLU mask): -5 instead of a backward that would read
    workspace bytes nobody wrote (advisor finding, round 5).  The check precedes every device access."""
    lib = _capi.lib()
    fake = ctypes.c_void_p(16)
    for prec in (1, 2):
        rc = lib.giga_encoder_forward(fake, fake, fake, None, 1, prec | _capi.CONVIN_MASK, fake, ctypes.c_size_t(1 << 40), None)
        assert rc == -5, (prec, rc)
