"""CPU check of the Winograd F(2x2, 3x3) kernel's data movement (giga_amd/csrc/giga_wino.h): a numpy emulation that reads the SAME
packed Winograd-domain weight image (giga_pack.cpp::pack_wino, the A operand of v_mfma_f32_16x16x4_f32) and follows the kernel lane by
lane -- tile blocks, haloed patches, the lane's own B^T d B on channel pairs, 16 positions x K-steps of the 16x16x4 MFMA (A[i = lane & 15]
[k = lane >> 4], B[k][j = lane & 15], D rows 4 (lane >> 4) + r), A^T M A per lane, K-passes for inputs wider than 64 channels --
must reproduce torch's conv2d + ReLU of every 3x3 U-Net layer (encoder/unet.py:14-23)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from giga_amd import _capi

# (c0, c1, cout, H) of the 3x3 layers in giga_layout.h::kConv order, with their index
LAYERS = {0: (32, 0, 32, 40), 1: (32, 0, 32, 40), 2: (32, 0, 64, 20), 3: (64, 0, 64, 20), 4: (64, 0, 128, 10), 5: (128, 0, 128, 10),
          7: (64, 64, 64, 20), 8: (64, 0, 64, 20), 10: (32, 32, 32, 40), 11: (32, 0, 32, 40)}
KEYS = {0: "down_convs.0.conv1", 1: "down_convs.0.conv2", 2: "down_convs.1.conv1", 3: "down_convs.1.conv2", 4: "down_convs.2.conv1",
        5: "down_convs.2.conv2", 7: "up_convs.0.conv1", 8: "up_convs.0.conv2", 10: "up_convs.1.conv1", 11: "up_convs.1.conv2"}


def wino_offsets(total):
    """byte offset of every layer's Winograd image: the last region of the blob before the 256-byte stamp (giga_layout.h)"""
    size = {l: 16 * (c0 + c1) * co * 4 for l, (c0, c1, co, _) in LAYERS.items()}
    at = total - 256 - sum(size.values())
    off = {}
    for l in sorted(LAYERS):
        off[l] = at
        at += size[l]
    return off


def mfma_16x16x4(A, B, C):
    """A, B: (64,) one float per lane; C: (64, 4).  D[i][j] = sum_k A[lane i + 16 k] * B[lane j + 16 k]; lane (j, g) register r = D[4 g + r][j]."""
    Am = A.reshape(4, 16)            # [k][i]
    Bm = B.reshape(4, 16)            # [k][j]
    D = np.einsum("ki,kj->ij", Am.astype(np.float64), Bm.astype(np.float64))
    out = C.astype(np.float64).copy()
    for g in range(4):
        for r in range(4):
            out[16 * g:16 * g + 16, r] += D[4 * g + r, :]
    return out.astype(np.float32)


def emulate_layer(blob, off, c0, c1, cout, H, x):
    """x: (nimg, H, H, cin) NHWC fp32 -> (nimg, H, H, cout) NHWC, pre-bias, as the kernel computes it"""
    W = H
    cin = c0 + c1
    KP = cin // 64 if cin > 64 else 1
    CINP = cin // KP
    NCHUNK = CINP // 16
    TW = TH = H // 2
    BW, BH = (4, 4) if TW % 4 == 0 else (5, 3)
    TXB, TYB = -(-TW // BW), -(-TH // BH)
    img = np.frombuffer(blob, dtype=np.float32, count=16 * cin * cout, offset=off).reshape(cout // 16, KP, 16, NCHUNK, 2, 64, 2)
    lane = np.arange(64)
    j, g = lane & 15, lane >> 4
    jt = np.where(j < BW * BH, j, 0)
    tyl, txl = jt // BW, jt % BW
    out = np.zeros((x.shape[0], H, W, cout), np.float32)
    xp = np.pad(x, ((0, 0), (1, 2 * BH), (1, 2 * BW), (0, 0)))          # zero halo (the kernel zero-fills out-of-image patch cells)
    for n in range(x.shape[0]):
        for by in range(TYB):
            for bx in range(TXB):
                ty, tx = BH * by + tyl, BW * bx + txl
                ok = (j < BW * BH) & (ty < TH) & (tx < TW)
                for grp in range(cout // 16):
                    y = np.zeros((4, 64, 4), np.float32)
                    for kp in range(KP):
                        acc = np.zeros((16, 64, 4), np.float32)
                        for cc in range(NCHUNK):
                            for h in range(2):
                                ch = kp * CINP + cc * 16 + 4 * g + 2 * h                  # this lane's channel pair
                                d = np.zeros((4, 4, 64, 2), np.float32)
                                for r in range(4):
                                    for c in range(4):
                                        for e in range(2):
                                            d[r, c, :, e] = xp[n, 2 * BH * by + 2 * tyl + r, 2 * BW * bx + 2 * txl + c, ch + e]
                                e_ = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])          # rows
                                v = np.stack([e_[:, 0] - e_[:, 2], e_[:, 1] + e_[:, 2], e_[:, 2] - e_[:, 1], e_[:, 1] - e_[:, 3]], axis=1)
                                for pos in range(16):
                                    for s in range(2):
                                        acc[pos] = mfma_16x16x4(img[grp, kp, pos, cc, h, :, s], v[pos >> 2, pos & 3, :, s], acc[pos])
                        t0 = [acc[4 * xi] + acc[4 * xi + 1] + acc[4 * xi + 2] for xi in range(4)]
                        t1 = [acc[4 * xi + 1] - acc[4 * xi + 2] - acc[4 * xi + 3] for xi in range(4)]
                        y[0] += t0[0] + t0[1] + t0[2]
                        y[1] += t1[0] + t1[1] + t1[2]
                        y[2] += t0[1] - t0[2] - t0[3]
                        y[3] += t1[1] - t1[2] - t1[3]
                    for e in range(4):
                        for l in np.nonzero(ok)[0]:
                            out[n, 2 * ty[l] + (e >> 1), 2 * tx[l] + (e & 1), grp * 16 + 4 * g[l]:grp * 16 + 4 * g[l] + 4] = y[e, l]
    return out


@pytest.mark.parametrize("layer", sorted(LAYERS))
def test_wino_emulation_matches_torch(sd7, layer):
    c0, c1, cout, H = LAYERS[layer]
    flat = torch.cat([v.reshape(-1) for v in sd7.values()])
    blob = _capi.pack_weights(flat, 15).numpy().tobytes()
    off = wino_offsets(len(blob))[layer]
    rng = np.random.default_rng(100 + layer)
    nimg = 1 if H == 40 or c0 + c1 > 64 else 2
    x = rng.standard_normal((nimg, H, H, c0 + c1)).astype(np.float32)
    got = emulate_layer(blob, off, c0, c1, cout, H, x)
    w = sd7[f"encoder.unet.{KEYS[layer]}.weight"]
    want = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).float().numpy()
    err = np.abs(got - want).max()
    assert err < 2e-5 * max(1.0, np.abs(want).max()), f"layer {layer}: {err:.3e}"
    # every output element written (random data: exact zeros do not occur)
    assert (got != 0).all()


@pytest.mark.parametrize("layer", [1, 3, 7, 10])
def test_wino_data_gradient_emulation_matches_autograd(sd7, layer):
    """The fp32 training step runs the 3x3 layers' DATA gradients on the same kernel with the Winograd images of the flipped / transposed
    weights that end the backward blob (giga_pack.cpp::pack_wino_dgrad; channels in = the layer's cout, out = its cin): the emulation on
    that image must reproduce autograd's dX of conv2d (scripts/train_giga.py:208 through encoder/unet.py:14-23)."""
    c0, c1, cout, H = LAYERS[layer]
    cin = c0 + c1
    flat = torch.cat([v.reshape(-1) for v in sd7.values()])
    blob = _capi.pack_bwd_weights(flat, 15).numpy().tobytes()
    off = wino_offsets(len(blob))[layer]                       # (same tail layout as the forward blob: images in layer order, then the stamp)
    rng = np.random.default_rng(200 + layer)
    dy = rng.standard_normal((1, H, H, cout)).astype(np.float32)
    got = emulate_layer(blob, off, cout, 0, cin, H, dy)        # a convolution with cout channels in, cin out
    w = sd7[f"encoder.unet.{KEYS[layer]}.weight"].double()
    x = torch.zeros((1, cin, H, H), dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    want = x.grad.permute(0, 2, 3, 1).float().numpy()
    err = np.abs(got - want).max()
    assert err < 2e-5 * max(1.0, np.abs(want).max()), f"layer {layer}: {err:.3e}"
