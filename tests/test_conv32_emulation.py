"""CPU check of conv32's data movement (giga_amd/csrc/giga_conv32.h, geometry in giga_conv32_geom.h): a byte-accurate emulation
(tests/emu/conv32_emu.cpp) that drives the SAME geometry functions as the gfx950 kernel and reads the SAME packed fragments must
reproduce every U-Net layer (encoder/unet.py:14-114,225-239) computed by torch with identically rounded operands -- for 1, 3 and 5
images per group (bands inside one image, bands across image boundaries, ragged splits), in the three arithmetic modes.  Every
output element must be written exactly once and no value may depend on an LDS byte the staging did not write (the emulated LDS is
poisoned with NaN patterns before every sub-band)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from giga_amd import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "conv32_emu.cpp")
CSRC = os.path.join(os.path.dirname(HERE), "giga_amd", "csrc")
OUT = os.path.join(HERE, "emu", "_build", "libconv32_emu.so")

# (kind, c0, c1, cout, H, W, poolin, state-dict key) in giga_layout.h::kConv order
LAYERS = [(0, 32, 0, 32, 40, 40, False, "down_convs.0.conv1"), (0, 32, 0, 32, 40, 40, False, "down_convs.0.conv2"),
          (0, 32, 0, 64, 20, 20, True, "down_convs.1.conv1"), (0, 64, 0, 64, 20, 20, False, "down_convs.1.conv2"),
          (0, 64, 0, 128, 10, 10, True, "down_convs.2.conv1"), (0, 128, 0, 128, 10, 10, False, "down_convs.2.conv2"),
          (1, 128, 0, 64, 10, 10, False, "up_convs.0.upconv"), (0, 64, 64, 64, 20, 20, False, "up_convs.0.conv1"),
          (0, 64, 0, 64, 20, 20, False, "up_convs.0.conv2"), (1, 64, 0, 32, 20, 20, False, "up_convs.1.upconv"),
          (0, 32, 32, 32, 40, 40, False, "up_convs.1.conv1"), (0, 32, 0, 32, 40, 40, False, "up_convs.1.conv2"),
          (2, 32, 0, 32, 40, 40, False, "conv_final")]


def _compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++", shutil.which("clang++") or ""):
        if c and os.path.exists(c):
            return c
    return None


@pytest.fixture(scope="module")
def emu():
    cxx = _compiler()
    if cxx is None:
        pytest.skip("no clang++ (the emulation uses _Float16)")
    deps = [SRC, os.path.join(CSRC, "giga_conv32_geom.h"), os.path.join(CSRC, "giga_layout.h")]
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run([cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, SRC, "-o", OUT], check=True)
    lib = ctypes.CDLL(OUT)
    lib.conv32_emu_layer.restype = ctypes.c_int
    lib.conv32_emu_layer.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
    return lib


def _operand(t, mode):
    """what the MFMA sees of a value: f16 (mode 0), hi + lo of the f16x3 split (mode 1: ~22 bits), bf16 (mode 2)"""
    if mode == 0:
        return t.half().float()
    if mode == 2:
        return t.bfloat16().float()
    hi = t.half().float()
    return hi + (t - hi).half().float()


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["f16", "f16x3", "bf16"])
@pytest.mark.parametrize("G", [1, 3, 5, 9])
def test_conv32_emulation_matches_torch(emu, sd7, mode, G):
    flat = torch.cat([v.reshape(-1) for v in sd7.values()])
    blob = _capi.pack_weights(flat, 15).numpy()
    rng = np.random.default_rng(100 + 10 * mode + G)
    for layer, (kind, c0, c1, cout, H, W, poolin, key) in enumerate(LAYERS):
        if G == 5 and layer not in (0, 2, 5, 6, 7, 12):      # (the large case on one layer of every kind / resolution)
            continue
        if G == 9 and layer not in (0, 3, 10):               # nine images per group: a member's band needs SEVERAL sub-bands
            continue                                         # (46 rows of a 40 x 40 layer against 40 / 14 that fit; 128-scene batches)
        ih, iw = (2 * H, 2 * W) if poolin else (H, W)
        oh, ow = (2 * H, 2 * W) if kind == 1 else (H, W)
        x0 = torch.from_numpy(rng.standard_normal((G, ih, iw, c0)).astype(np.float32))
        x1 = torch.from_numpy(rng.standard_normal((G, ih, iw, max(c1, 1))).astype(np.float32))
        if mode == 0:                                        # native f16 activations live in memory as f16
            x0, x1 = x0.half().float(), x1.half().float()
        out = np.full((G, oh, ow, cout), np.nan, np.float32)
        pool = np.full((G, H, W, c0), np.nan, np.float32)
        written = np.zeros(out.shape, np.int32)
        stats = np.zeros(4, np.int32)
        a0, a1 = np.ascontiguousarray(x0.numpy()), np.ascontiguousarray(x1.numpy())
        rc = emu.conv32_emu_layer(layer, mode, blob.ctypes.data, G, a0.ctypes.data, a1.ctypes.data if c1 else None,
                                  out.ctypes.data, pool.ctypes.data if poolin else None, written.ctypes.data, stats.ctypes.data)
        assert rc == 0, (layer, rc)
        assert int(written.min()) == 1 and int(written.max()) == 1, (layer, "outputs written", int(written.min()), int(written.max()))
        assert np.isfinite(out).all(), (layer, "a valid output read an LDS byte that was never staged")
        assert stats[0] <= 160 * 1024 - 1024
        if G == 9 and layer in (0, 10):
            assert 9 * 41 // 8 > stats[2], (layer, "expected more rows per member than one sub-band holds", int(stats[2]))
        # torch with the same operand rounding
        w = _operand(sd7[f"encoder.unet.{key}.weight"], mode)
        bias = sd7[f"encoder.unet.{key}.bias"]
        xin = torch.cat((x0, x1), 3) if c1 else x0
        xin = xin.permute(0, 3, 1, 2)
        if poolin:
            xin = F.max_pool2d(xin, 2, 2)
            assert np.array_equal(pool, xin.permute(0, 2, 3, 1).numpy()), (layer, "pooled write-through")
        xin = _operand(xin, mode)
        if kind == 0:
            ref = F.relu(F.conv2d(xin.double(), w.double(), bias.double(), padding=1))
        elif kind == 1:
            ref = F.conv_transpose2d(xin.double(), w.double(), bias.double(), stride=2)
        else:
            ref = F.conv2d(xin.double(), w.double(), bias.double())
        ref = ref.permute(0, 2, 3, 1).float()
        got = torch.from_numpy(out)
        scale = max(1.0, float(ref.abs().max()))
        if mode == 0:                                        # the emulation rounds its outputs to f16 like the kernel
            err = float((got - ref.half().float()).abs().max())
            assert err <= scale * 2.0 ** -10, (layer, err)
            assert float(((got - ref.half().float()).abs() > 0).float().mean()) < 0.01
        else:
            tol = 3e-6 if mode == 2 else 2e-5                # bf16: exact products; f16x3: the dropped lo x lo term, 2^-22 relative
            assert float((got - ref).abs().max()) <= tol * scale, (layer, float((got - ref).abs().max()), scale)


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["f16", "f16x3", "bf16"])
@pytest.mark.parametrize("G", [1, 3, 5])
def test_conv32_fused_pair_emulation_equals_two_layers(emu, sd7, mode, G):
    """The fused pairs of the persistent kernel (giga_conv32_geom.h: C32Pair -- layers (0,1), (2,3), (10,11): the second layer
    reads the first one's output from LDS, the member recomputes one row above and below its band) must give the SAME bits as
    the two layers run one after the other, write every output of both layers exactly once, and fill every byte of the second
    layer's LDS image exactly once.  Modes whose weights do not fit side by side report 1 (the kernel keeps the barrier there)."""
    emu.conv32_emu_pair.restype = ctypes.c_int
    emu.conv32_emu_pair.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8
    flat = torch.cat([v.reshape(-1) for v in sd7.values()])
    blob = _capi.pack_weights(flat, 15).numpy()
    rng = np.random.default_rng(700 + 10 * mode + G)
    fused = 0
    for la in (0, 2, 10):
        kind, c0, c1, cout, H, W, poolin, _ = LAYERS[la]
        ih, iw = (2 * H, 2 * W) if poolin else (H, W)
        x0 = rng.standard_normal((G, ih, iw, c0)).astype(np.float32)
        x1 = rng.standard_normal((G, ih, iw, max(c1, 1))).astype(np.float32)
        if mode == 0:
            x0, x1 = x0.astype(np.float16).astype(np.float32), x1.astype(np.float16).astype(np.float32)
        cb = LAYERS[la + 1][3]
        outA, outB = np.full((G, H, W, cout), np.nan, np.float32), np.full((G, H, W, cb), np.nan, np.float32)
        wA, wB = np.zeros(outA.shape, np.int32), np.zeros(outB.shape, np.int32)
        pool = np.full((G, H, W, c0), np.nan, np.float32)
        stats = np.zeros(4, np.int32)
        rc = emu.conv32_emu_pair(la, mode, blob.ctypes.data, G, x0.ctypes.data, x1.ctypes.data if c1 else None, outA.ctypes.data,
                                 pool.ctypes.data if poolin else None, wA.ctypes.data, outB.ctypes.data, wB.ctypes.data, stats.ctypes.data)
        if rc == 1:
            continue
        assert rc == 0, (la, rc)
        fused += 1
        assert wA.min() == 1 and wA.max() == 1 and wB.min() == 1 and wB.max() == 1, (la, wA.min(), wA.max(), wB.min(), wB.max())
        assert stats[0] <= 160 * 1024 - 1024 and stats[2] >= 6
        # the two layers separately
        uA, uB = np.full(outA.shape, np.nan, np.float32), np.full(outB.shape, np.nan, np.float32)
        pool2 = np.full(pool.shape, np.nan, np.float32)
        w2 = np.zeros(outA.shape, np.int32)
        assert emu.conv32_emu_layer(la, mode, blob.ctypes.data, G, x0.ctypes.data, x1.ctypes.data if c1 else None, uA.ctypes.data,
                                    pool2.ctypes.data if poolin else None, w2.ctypes.data, stats.ctypes.data) == 0
        w3 = np.zeros(outB.shape, np.int32)
        assert emu.conv32_emu_layer(la + 1, mode, blob.ctypes.data, G, uA.ctypes.data, None, uB.ctypes.data, None, w3.ctypes.data,
                                    stats.ctypes.data) == 0
        assert np.array_equal(outA, uA), (la, "first layer of the pair")
        assert np.array_equal(outB, uB), (la, "second layer of the pair")
        if poolin:
            assert np.array_equal(pool, pool2)
    assert fused == (3 if mode != 1 else 0)


def test_conv_in_f16_slot_table(emu):
    """The K-slot order of the f16-class conv_in (giga_layout.h: ci16_tap / ci16_read, shared by the packer and
    convin_project_kernel<.., SPLIT>): every one of the 27 taps carries its weight in exactly one slot, a slot without a weight
    still reads a voxel inside the staged sub-volume, and in a 32-bank model of the LDS every ds_read_b32 of the gather -- 32 lanes
    of a half-wave = 16 voxels (2 iy x 8 iz) x the two k-groups (0, 1) or (2, 3) -- touches 32 different banks."""
    tap, read, strides = np.zeros(32, np.int32), np.zeros(32, np.int32), np.zeros(2, np.int32)
    emu.ci16_table.restype = None
    emu.ci16_table.argtypes = [ctypes.c_void_p] * 3
    emu.ci16_table(tap.ctypes.data, read.ctypes.data, strides.ctypes.data)
    RS, SLAB = int(strides[0]), int(strides[1])
    assert sorted(t for t in tap if t >= 0) == list(range(27))
    assert ((read >= 0) & (read < 27)).all() and all(read[i] == tap[i] for i in range(32) if tap[i] >= 0)
    assert RS >= 45 and SLAB >= 12 * RS                      # 42 values + the alignment offset per row, 12 rows per slab
    for e in range(8):
        for half in range(2):
            banks = {}
            for lane in range(32 * half, 32 * half + 32):
                j, g = lane & 15, lane >> 4
                t = int(read[8 * g + e])
                a = (t // 9) * SLAB + ((j >> 3) + (t // 3) % 3) * RS + 3 + 4 * ((j >> 2) & 1) + (j & 3) + t % 3
                banks.setdefault(a % 32, set()).add(a)
            assert max(len(v) for v in banks.values()) == 1, (e, half)
