"""CPU check of the bf16 training decoder's weight images (csrc/giga_dect.h): a numpy emulation of the kernel's MFMA chains
(forward out of the forward image, gradient chain out of the transposed image; fragment order and k-slot maps of
csrc/giga_decoder_train16.hip) must reproduce the operand-rounded reference tests/dect_ref.py.  Catches packer / kernel
convention mismatches without a GPU."""
import numpy as np
import torch

from giga_amd import _capi, weights
from tests import dect_ref as R
from tests.mfma_emu import drow, mfma

HEADS = weights.HEADS
OUT_DIM = {"decoder_qual": 1, "decoder_rot": 4, "decoder_width": 1, "decoder_tsdf": 1}
FWD_BYTES, BWD_BYTES = 59 * 1024, 51 * 1024


def _bf_bits_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _bf(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def _images(sd):
    flat = torch.cat([v.reshape(-1) for v in sd.values()])
    blob = _capi.pack_weights(flat, 15).numpy()
    bblob = _capi.pack_bwd_weights(flat, 15).numpy()
    up = lambda x: (x + 255) // 256 * 256  # noqa: E731
    wino = 16 * 48128 * 4           # round 6: the Winograd images of the ten 3x3 layers (sum of cin * cout = 48 128) sit between them and the stamp
    fwd = [blob[blob.size - 256 - wino - (4 - h) * up(FWD_BYTES):][:FWD_BYTES] for h in range(4)]   # the last regions of both blobs (before the stamp)
    bwd = [bblob[bblob.size - 256 - wino - (4 - h) * BWD_BYTES:][:BWD_BYTES] for h in range(4)]
    return fwd, bwd


def _chain_operand(x32, ch):
    """bf16 B operand of chunk ch from a (32 points, 32 features) array in D-register order: slot (hi, j) = feature drow(8ch + j, hi)."""
    out = np.zeros((64, 8), np.float32)
    for hi in range(2):
        for j in range(8):
            out[hi * 32:(hi + 1) * 32, j] = x32[:, drow(8 * ch + j, hi)]
    return _bf(out)


def _regs_to_feat(d):
    """(64, 16) D registers -> (32 points, 32 features)."""
    out = np.zeros((32, 32), np.float32)
    for hi in range(2):
        for r in range(16):
            out[:, drow(r, hi)] = d[hi * 32:(hi + 1) * 32, r]
    return out


def _feat_to_regs(x):
    d = np.zeros((64, 16), np.float32)
    for hi in range(2):
        for r in range(16):
            d[hi * 32:(hi + 1) * 32, r] = x[:, drow(r, hi)]
    return d


def test_bf16_training_images_forward_and_gradient_chain(sd7):
    fwd, bwd = _images(sd7)
    rng = np.random.default_rng(21)
    c = rng.standard_normal((32, 96)).astype(np.float32)
    p = rng.random((32, 3)).astype(np.float32) - 0.5
    for h, name in enumerate(HEADS):
        W = _bf_bits_to_f32(fwd[h][:58 * 1024].view(np.uint16)).reshape(58, 64, 8)
        ctab = fwd[h][58 * 1024:58 * 1024 + 768].view(np.float32)
        WB = _bf_bits_to_f32(bwd[h][:50 * 1024].view(np.uint16)).reshape(50, 64, 8)
        wout = bwd[h][50 * 1024:50 * 1024 + 512].view(np.float32).reshape(4, 32)
        cb = _bf(c)
        cf = np.zeros((6, 64, 8), np.float32)
        for ch in range(6):
            for hi in range(2):
                for j in range(8):
                    cf[ch, hi * 32:(hi + 1) * 32, j] = cb[:, (ch // 2) * 32 + (ch % 2) * 16 + 8 * hi + j]
        ph = _bf(p)
        plo = _bf(p - ph)
        ax = np.zeros((64, 8), np.float32)
        ax[:32, 0:3] = ph; ax[:32, 3] = 1; ax[:32, 4:7] = plo; ax[:32, 7] = 1
        ax[32:, 0:3] = ph

        def ctab_regs(blk):
            d = np.zeros((64, 16), np.float32)
            for hi in range(2):
                for r in range(16):
                    d[hi * 32:(hi + 1) * 32, r] = ctab[blk * 32 + drow(r, hi)]
            return d

        relu_bf = lambda d, ch: _bf(np.maximum(d[:, 8 * ch:8 * ch + 8], 0))  # noqa: E731
        # ---- forward, the kernel's order
        net = np.zeros((64, 16), np.float32)
        for ch in range(7):
            net = mfma(W[ch], cf[ch] if ch < 6 else ax, net)
        XN, XH = [], []
        for blk in range(5):
            k = 11 * blk
            XN.append([relu_bf(net, 0), relu_bf(net, 1)])
            hh = mfma(W[k + 7], XN[blk][0], ctab_regs(blk))
            hh = mfma(W[k + 8], XN[blk][1], hh)
            if blk < 4:
                for ch in range(7):
                    net = mfma(W[k + 11 + ch], cf[ch] if ch < 6 else ax, net)
            else:
                net = mfma(W[55], ax, net)
            XH.append([relu_bf(hh, 0), relu_bf(hh, 1)])
            net = mfma(W[k + 9], XH[blk][0], net)
            net = mfma(W[k + 10], XH[blk][1], net)
        XO = [relu_bf(net, 0), relu_bf(net, 1)]
        o = mfma(W[56], XO[0], ctab_regs(5))
        o = mfma(W[57], XO[1], o)
        got = o[:32, :OUT_DIM[name]]
        ref, saved = R.head_forward(sd7, name + ".", torch.from_numpy(c), torch.from_numpy(p))
        scale = max(1.0, float(ref.abs().max()))
        # accumulation order only -- except where a 1e-7 difference tips a bf16 rounding of some activation (one part in 2^9 of ONE
        # operand): nearly every point agrees to 2e-5, none is off by more than such flips explain
        err = np.abs(got - ref.numpy())
        assert (err > 2e-5 * scale).mean() <= 0.1 and err.max() < 5e-3 * scale, (name, err.max(), (err > 2e-5 * scale).mean())
        # the activations the backward keeps: XN / XH operands == the reference's rounded activations
        for blk in range(5):
            xn = np.concatenate([XN[blk][0], XN[blk][1]], axis=1)          # (64, 16) in D-register order
            assert np.abs(_regs_to_feat(xn) - saved["xn"][blk].numpy()).max() < 2e-2 * scale  # (rounding-boundary flips only)
        # ---- gradient chain
        dO = rng.standard_normal((32, OUT_DIM[name])).astype(np.float32)
        xo_feat = _regs_to_feat(np.concatenate(XO, axis=1))
        G = (dO @ wout[:OUT_DIM[name]]) * (xo_feat != 0)
        dc = [np.zeros((64, 16), np.float32) for _ in range(3)]
        z = np.zeros((64, 16), np.float32)
        for blk in range(4, -1, -1):
            kb = 10 * blk
            Gb = [_chain_operand(G, 0), _chain_operand(G, 1)]
            dh = mfma(WB[kb + 9], Gb[1], mfma(WB[kb + 8], Gb[0], z))
            xh_feat = _regs_to_feat(np.concatenate(XH[blk], axis=1))
            DH = _regs_to_feat(dh) * (xh_feat != 0)
            Hb = [_chain_operand(DH, 0), _chain_operand(DH, 1)]
            dn = mfma(WB[kb + 7], Hb[1], mfma(WB[kb + 6], Hb[0], z))
            xn_feat = _regs_to_feat(np.concatenate(XN[blk], axis=1))
            G = G + _regs_to_feat(dn) * (xn_feat != 0)
            Gb = [_chain_operand(G, 0), _chain_operand(G, 1)]
            for pl in range(3):
                dc[pl] = mfma(WB[kb + 2 * pl + 1], Gb[1], mfma(WB[kb + 2 * pl], Gb[0], dc[pl]))
        dc_feat = np.concatenate([_regs_to_feat(d) for d in dc], axis=1)   # (32, 96)
        # reference from ITS OWN saved activations; the emulation's differ by rounding-boundary flips at most, so compare loosely
        # in value but exactly in structure: a transposed / permuted fragment would be off by O(1)
        _, dc_ref = R.head_backward(sd7, name + ".", saved, torch.from_numpy(dO))
        s = max(1e-6, float(dc_ref.abs().max()))
        assert np.abs(dc_feat - dc_ref.numpy()).max() < 2e-2 * s, (name, np.abs(dc_feat - dc_ref.numpy()).max(), s)
        assert np.linalg.norm(dc_feat - dc_ref.numpy()) < 2e-3 * np.linalg.norm(dc_ref.numpy()), name


def test_tile_ownership_covers_every_gradient_tile_once():
    """wave w plays role (w + b) & 3 in block b (giga_decoder_train16.hip): every one of the 31 tiles has exactly one owner and no
    wave holds more than 8 accumulator tiles."""
    ntiles = lambda role: 2 if role < 2 else 1  # noqa: E731
    first = {0: 0, 1: 2, 2: 4, 3: 5}
    owned = {}
    per_wave = [0, 0, 0, 0]
    for w in range(4):
        for b in range(5):
            role = (w + b) & 3
            for u in range(ntiles(role)):
                t = 6 * b + first[role] + u
                assert t not in owned
                owned[t] = w
                per_wave[w] += 1
    owned[30] = 2; per_wave[2] += 1
    assert sorted(owned) == list(range(31)) and max(per_wave) == 8, per_wave
