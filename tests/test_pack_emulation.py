"""CPU check of the packed-weight fragment conventions: a numpy emulation of the decoder kernels'
MFMA chains (same fragment order, same k-slot maps as giga_decoder.hip) must reproduce the oracle.
Catches packer/kernel convention mismatches without a GPU."""
import numpy as np
import torch

from giga_amd import _capi, weights
from oracle import giga_oracle as O
from tests.mfma_emu import mfma

HEADS = weights.HEADS
OUT_DIM = {"decoder_qual": 1, "decoder_rot": 4, "decoder_width": 1, "decoder_tsdf": 1}


def _blob_and_offsets(sd):
    flat = torch.cat([v.reshape(-1) for v in sd.values()])
    blob = _capi.pack_weights(flat, 15).numpy()
    # mirror giga_layout.h::pack_offsets()
    def up(x, a=256):
        return (x + a - 1) // a * a
    at = 14 * 64 * 4 + up(32 * 4)
    conv = [(0, 32, 0, 32), (0, 32, 0, 32), (0, 32, 0, 64), (0, 64, 0, 64), (0, 64, 0, 128), (0, 128, 0, 128),
            (1, 128, 0, 64), (0, 64, 64, 64), (0, 64, 0, 64), (1, 64, 0, 32), (0, 32, 32, 32), (0, 32, 0, 32),
            (2, 32, 0, 32)]
    tail = 0
    for kind, c0, c1, co in conv:
        cin = c0 + c1
        nblk = co // 32 * (4 if kind == 1 else 1)
        taps = 9 if kind == 0 else 1
        at += nblk * taps * (cin // 16) * 1024 + nblk * taps * (cin // 8) * 1024 + up(co * 4)
        tail += 2 * nblk * taps * (cin // 16) * 1024          # f16x3 split conv fragments ([hi, lo] pairs), appended
    tail += 4 * 1024                                           # f16x3 split conv_in fragments
    tail += sum((co // 32 * (4 if kind == 1 else 1)) * (9 if kind == 0 else 1) * ((c0 + c1) // 16) * 1024
                for kind, c0, c1, co in conv)                  # bf16 conv fragments (f16 fragment layout)
    dec16, dec32 = [], []
    for _ in range(4):
        dec16.append(at); at += up(59 * 1024)          # 58 fragments + C table in a 59th 1 KiB chunk
        dec32.append(at); at += up(111 * 1024)
    dec16f, dec32f = [], []                            # the same heads with conv_final folded into fc_c (GIGA_FOLD_FINAL)
    for _ in range(4):
        dec16f.append(at); at += up(59 * 1024)
        dec32f.append(at); at += up(111 * 1024)
    dec16s = []                                        # f16x3 split images (plain, folded), appended
    for _ in range(4):
        dec16s.append(at); at += 2 * up(111 * 1024)
    at += tail
    # conv32 images (round 4): f16 + f16x3 [hi, lo] pairs + bf16 = 4 fragments per (32-channel slice, tap, 16-channel chunk)
    at += sum(4 * (co // 32 * (4 if kind == 1 else 1)) * (9 if kind == 0 else 1) * ((c0 + c1) // 16) * 1024 for kind, c0, c1, co in conv)
    at += 4 * up(59 * 1024)                            # bf16 images of the bf16 training decoder (round 5), appended
    at += sum(16 * (c0 + c1) * co * 4 for kind, c0, c1, co in conv if kind == 0)    # Winograd images of the 3x3 layers (round 6, giga_wino.h)
    at += 256                                          # the stamp (giga_packed_check)
    assert at == blob.size
    _blob_and_offsets.dec16s = dec16s
    return blob, dec16, dec32


def _ctab_regs(ctab, blk):
    c = np.zeros((64, 16), np.float32)
    for hi in range(2):
        for r in range(16):
            c[hi * 32:(hi + 1) * 32, r] = ctab[blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]
    return c


def _inputs(seed=3):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((1, 32, 96)).astype(np.float32)
    p = (rng.random((1, 32, 3)).astype(np.float32) - 0.5)
    return c, p


def test_decoder_f16_fragment_chain(sd7):
    blob, dec16, _ = _blob_and_offsets(sd7)
    c, p = _inputs()
    for h, name in enumerate(HEADS):
        raw = blob[dec16[h]:dec16[h] + 58 * 1024 + 768]
        W = raw[:58 * 1024].view(np.float16).reshape(58, 64, 8)
        ctab = raw[58 * 1024:].view(np.float32)
        # B operands as the kernel builds them
        cf = np.zeros((6, 64, 8), np.float16)
        for ch in range(6):
            for hi in range(2):
                for j in range(8):
                    cf[ch, hi * 32:(hi + 1) * 32, j] = c[0, :, (ch // 2) * 32 + (ch % 2) * 16 + 8 * hi + j]
        ph = p[0].astype(np.float16)
        plo = (p[0] - ph.astype(np.float32)).astype(np.float16)
        ax = np.zeros((64, 8), np.float16)
        ax[:32, 0:3] = ph; ax[:32, 3] = 1; ax[:32, 4:7] = plo; ax[:32, 7] = 1
        ax[32:, 0:3] = ph

        def pack_relu(d, ch):
            return np.maximum(d[:, 8 * ch:8 * ch + 8], 0).astype(np.float16)

        net = np.zeros((64, 16), np.float32)
        k = 0
        for blk in range(5):
            for ch in range(7):
                net = mfma(W[k], cf[ch] if ch < 6 else ax, net); k += 1
            hh = mfma(W[k], pack_relu(net, 0), _ctab_regs(ctab, blk))
            hh = mfma(W[k + 1], pack_relu(net, 1), hh); k += 2
            net = mfma(W[k], pack_relu(hh, 0), net)
            net = mfma(W[k + 1], pack_relu(hh, 1), net); k += 2
        net = mfma(W[k], ax, net); k += 1
        o = mfma(W[k], pack_relu(net, 0), _ctab_regs(ctab, 5))
        o = mfma(W[k + 1], pack_relu(net, 1), o)
        got = o[:32, :OUT_DIM[name]]                       # hi = 0 lanes, registers 0..3 = rows 0..3
        ref = O.decoder_mlp(sd7, name, torch.from_numpy(p), torch.from_numpy(c)).numpy().reshape(32, -1)
        err = np.abs(got - ref).max()
        assert err < 5e-3, (name, err)                     # f16 operand rounding only


def test_decoder_f32_fragment_chain(sd7):
    blob, _, dec32 = _blob_and_offsets(sd7)
    c, p = _inputs(5)
    for h, name in enumerate(HEADS):
        raw = blob[dec32[h]:dec32[h] + 110 * 1024 + 768]
        W = raw[:110 * 1024].view(np.float32).reshape(110, 64, 4)
        ctab = raw[110 * 1024:].view(np.float32)

        def one(Wk, j, bvec, acc):     # one 32x32x2 MFMA: A = component j of fragment, B = per-lane scalar
            return mfma(Wk[:, j:j + 1], bvec.reshape(64, 1), acc)

        cf = np.zeros((48, 64), np.float32)
        for m in range(48):
            for hi in range(2):
                cf[m, hi * 32:(hi + 1) * 32] = c[0, :, (m // 16) * 32 + 16 * hi + (m % 16)]
        ax0 = np.concatenate([p[0, :, 0], p[0, :, 1]])
        ax1 = np.concatenate([p[0, :, 2], np.ones(32, np.float32)])
        ax2 = np.concatenate([np.ones(32, np.float32), np.zeros(32, np.float32)])
        net = np.zeros((64, 16), np.float32)
        k = 0
        for blk in range(5):
            for q in range(12):
                for j in range(4):
                    net = one(W[k], j, cf[4 * q + j], net)
                k += 1
            net = one(W[k], 0, ax0, net); net = one(W[k], 1, ax1, net); net = one(W[k], 2, ax2, net); k += 1
            hh = _ctab_regs(ctab, blk)
            for q in range(4):
                for j in range(4):
                    hh = one(W[k], j, np.maximum(net[:, 4 * q + j], 0), hh)
                k += 1
            new = net.copy()
            for q in range(4):
                for j in range(4):
                    new = one(W[k], j, np.maximum(hh[:, 4 * q + j], 0), new)
                k += 1
            net = new
        net = one(W[k], 1, ax1, net); k += 1
        o = _ctab_regs(ctab, 5)
        for q in range(4):
            for j in range(4):
                o = one(W[k], j, np.maximum(net[:, 4 * q + j], 0), o)
            k += 1
        assert k == 110
        got = o[:32, :OUT_DIM[name]]
        ref = O.decoder_mlp(sd7, name, torch.from_numpy(p), torch.from_numpy(c)).numpy().reshape(32, -1)
        err = np.abs(got - ref).max()
        assert err < 2e-5, (name, err)


def test_decoder_f16x3_split_fragment_chain(sd7):
    """The split image [hi, lo] pairs + the kernel's product order W_lo*x_hi + W_hi*x_lo + W_hi*x_hi
    (decoder_f16s_kernel) reproduce the fp32 oracle to fp32-rounding level."""
    blob, _, _ = _blob_and_offsets(sd7)
    dec16s = _blob_and_offsets.dec16s
    c, p = _inputs(11)

    def split(x):
        h = x.astype(np.float16)
        return h, (x - h.astype(np.float32)).astype(np.float16)

    for h, name in enumerate(HEADS):
        raw = blob[dec16s[h]:dec16s[h] + 110 * 1024 + 768]
        W = raw[:110 * 1024].view(np.float16).reshape(110, 64, 8)
        ctab = raw[110 * 1024:].view(np.float32)
        cf = np.zeros((6, 64, 8), np.float32)
        for ch in range(6):
            for hi in range(2):
                for j in range(8):
                    cf[ch, hi * 32:(hi + 1) * 32, j] = c[0, :, (ch // 2) * 32 + (ch % 2) * 16 + 8 * hi + j]
        cfh, cfl = split(cf)
        ph = p[0].astype(np.float16)
        plo = (p[0] - ph.astype(np.float32)).astype(np.float16)
        ax = np.zeros((64, 8), np.float16)
        ax[:32, 0:3] = ph; ax[:32, 3] = 1; ax[:32, 4:7] = plo; ax[:32, 7] = 1
        ax[32:, 0:3] = ph

        def mm3(f, xh, xl, acc):
            acc = mfma(W[f + 1], xh, acc)
            acc = mfma(W[f], xl, acc)
            return mfma(W[f], xh, acc)

        def dense(f0, src, dst):
            for ch in range(2):
                xh, xl = split(np.maximum(src[:, 8 * ch:8 * ch + 8], 0))
                dst = mm3(f0 + 2 * ch, xh, xl, dst)
            return dst

        def fc_c(blk, net):
            for ch in range(6):
                net = mm3(21 * blk + 2 * ch, cfh[ch], cfl[ch], net)
            return mfma(W[21 * blk + 12], ax, net)

        net = fc_c(0, np.zeros((64, 16), np.float32))
        for blk in range(5):
            hh = dense(21 * blk + 13, net, _ctab_regs(ctab, blk))
            net = fc_c(blk + 1, net) if blk < 4 else mfma(W[105], ax, net)
            net = dense(21 * blk + 17, hh, net)
        o = dense(106, net, _ctab_regs(ctab, 5))
        got = o[:32, :OUT_DIM[name]]
        ref = O.decoder_mlp(sd7, name, torch.from_numpy(p), torch.from_numpy(c)).numpy().reshape(32, -1)
        err = np.abs(got - ref).max()
        assert err < 2e-5, (name, err)
