// Shared host-side description of the GIGA network and of the packed-weight blob.
//
// Reference architecture (fixed by vgn.networks.GIGA, /root/reference/src/vgn/networks.py:91-115):
//   encoder  LocalVoxelEncoder: Conv3d(1,32,3,pad=1)+ReLU -> 3 axis-mean planes (40x40x32) ->
//            shared UNet(depth 3, start_filts 32, concat merge) -> 1x1 conv   (encoder/voxels.py, unet.py)
//   decoders 4x LocalDecoder(c_dim=3*32, hidden 32, 5 ResnetBlockFC, out_dim 1/4/1/1) (models/decoder.py)
//
// The flat fp32 parameter buffer handed to giga_pack_weights() is the reference state-dict
// flattened in its own key order (decoder_qual, decoder_rot, decoder_width, [decoder_tsdf], encoder).
#pragma once
#include <cstddef>
#include <cstdint>

namespace giga {

constexpr int RES = 40;            // TSDF / plane resolution
constexpr int CD = 32;             // c_dim == hidden_size
constexpr int NBLK = 5;            // ResnetBlockFC blocks per head
constexpr int NHEADS = 4;          // qual, rot, width, tsdf
constexpr int HEAD_OUT[NHEADS] = {1, 4, 1, 1};

// ----- flat fp32 parameter offsets (state-dict order) ---------------------------------------
struct HeadParamOff {
    size_t fc_c_w[NBLK], fc_c_b[NBLK];      // (32,96), (32)
    size_t fc_p_w, fc_p_b;                  // (32,3), (32)
    size_t fc0_w[NBLK], fc0_b[NBLK];        // (32,32), (32)
    size_t fc1_w[NBLK], fc1_b[NBLK];
    size_t out_w, out_b;                    // (out,32), (out)
};

enum ConvKind { CONV3 = 0, UPCONV = 1, CONV1 = 2, DOWN = 3 };   // DOWN: data gradient of UPCONV (2x2, stride 2)
struct ConvLayerDesc {
    int kind, cin0, cin1, cout, H, W;       // H,W = INPUT spatial size
    bool pool;                              // fused 2x2 max-pool output (DownConv with pooling)
};
constexpr int NCONV = 13;
constexpr ConvLayerDesc kConv[NCONV] = {
    {CONV3, 32, 0, 32, 40, 40, false},   // 0  down_convs.0.conv1
    {CONV3, 32, 0, 32, 40, 40, true},    // 1  down_convs.0.conv2 (+pool)
    {CONV3, 32, 0, 64, 20, 20, false},   // 2  down_convs.1.conv1
    {CONV3, 64, 0, 64, 20, 20, true},    // 3  down_convs.1.conv2 (+pool)
    {CONV3, 64, 0, 128, 10, 10, false},  // 4  down_convs.2.conv1
    {CONV3, 128, 0, 128, 10, 10, false}, // 5  down_convs.2.conv2 (no pool, unet.py:190)
    {UPCONV, 128, 0, 64, 10, 10, false}, // 6  up_convs.0.upconv  ConvTranspose2d(128,64,2,2)
    {CONV3, 64, 64, 64, 20, 20, false},  // 7  up_convs.0.conv1   cat(from_up, from_down)
    {CONV3, 64, 0, 64, 20, 20, false},   // 8  up_convs.0.conv2
    {UPCONV, 64, 0, 32, 20, 20, false},  // 9  up_convs.1.upconv
    {CONV3, 32, 32, 32, 40, 40, false},  // 10 up_convs.1.conv1
    {CONV3, 32, 0, 32, 40, 40, false},   // 11 up_convs.1.conv2
    {CONV1, 32, 0, 32, 40, 40, false},   // 12 conv_final (1x1, no activation, unet.py:238)
};
// order of the conv layers inside the state dict (down0.c1,c2, down1.., down2.., up0.upconv,c1,c2, up1.., final)
// is identical to kConv order.

struct ParamOff {
    HeadParamOff head[NHEADS];
    size_t conv_in_w, conv_in_b;            // (32,1,3,3,3), (32)
    size_t conv_w[NCONV], conv_b[NCONV];
    size_t total;
};

inline int conv_taps(const ConvLayerDesc& d) { return d.kind == CONV3 ? 9 : 1; }
inline int conv_nsub(const ConvLayerDesc& d) { return d.kind == UPCONV ? 4 : 1; }
inline size_t conv_w_count(const ConvLayerDesc& d) {
    return (size_t)d.cout * (d.cin0 + d.cin1) * (d.kind == CONV3 ? 9 : d.kind == UPCONV ? 4 : 1);
}

// head_present: bit h set <=> head h's parameters are present in the flat buffer.
inline ParamOff param_offsets(int head_present) {
    ParamOff o{};
    size_t at = 0;
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_present >> h & 1)) continue;
        HeadParamOff& p = o.head[h];
        for (int i = 0; i < NBLK; ++i) { p.fc_c_w[i] = at; at += CD * 3 * CD; p.fc_c_b[i] = at; at += CD; }
        p.fc_p_w = at; at += CD * 3; p.fc_p_b = at; at += CD;
        for (int i = 0; i < NBLK; ++i) {
            p.fc0_w[i] = at; at += CD * CD; p.fc0_b[i] = at; at += CD;
            p.fc1_w[i] = at; at += CD * CD; p.fc1_b[i] = at; at += CD;
        }
        p.out_w = at; at += HEAD_OUT[h] * CD; p.out_b = at; at += HEAD_OUT[h];
    }
    o.conv_in_w = at; at += CD * 27; o.conv_in_b = at; at += CD;
    for (int l = 0; l < NCONV; ++l) {
        o.conv_w[l] = at; at += conv_w_count(kConv[l]);
        o.conv_b[l] = at; at += kConv[l].cout;
    }
    o.total = at;
    return o;
}

// ----- packed blob ---------------------------------------------------------------------------
// A "fragment" is the 1 KiB MFMA operand image of one wave: 64 lanes x 16 bytes, lane-linear, so a
// wave reads it with one ds_read_b128 / global_load_dwordx4 per lane (conflict-free, coalesced).
//   f16 fragment : lane (n = lane&31, hi = lane>>5) holds 8 halfs = k-slots (hi, j=0..7) of one
//                  v_mfma_f32_32x32x16_f16.
//   f32 fragment : lane holds 4 floats = its k-slot `hi` of FOUR consecutive v_mfma_f32_32x32x2_f32.
constexpr size_t FRAG = 1024;

// Decoder head blob, precision f16: fragments in order of use
//   for blk 0..4: 6 feature frags + 1 aux frag (fc_c[blk]; aux = fc_p hi/lo (blk 0) + folded biases),
//                 2 frags fc_0, 2 frags fc_1
//   1 aux frag (b1 of block 4), 2 frags fc_out (rows >= out_dim zero)
//   then the fp32 C-init table: 5 x 32 (fc_0 bias) + 32 (fc_out bias, zero padded)
constexpr int DEC16_FRAGS = NBLK * (7 + 2 + 2) + 1 + 2;                   // 58
constexpr size_t DEC_CTAB_BYTES = (NBLK + 1) * CD * sizeof(float);        // 768
constexpr size_t DEC16_BYTES = (DEC16_FRAGS + 1) * FRAG;                  // 59 KiB: 58 fragments + the C table in a
                                                                          // 59th 1 KiB chunk (whole image = 59 LDS-DMA wave-chunks)
// precision f16x3 ("split": every operand is a pair hi = f16(v), lo = f16(v - hi); products hi*hi + hi*lo + lo*hi with fp32
// accumulation, i.e. ~22-bit operands at 3x the f16 MFMA count).  Fragments in order of use, [hi, lo] pairs:
//   for blk 0..4: 6 x [hi, lo] feature frags, 1 aux frag (already exact: it carries its own hi/lo slots),
//                 2 x [hi, lo] fc_0, 2 x [hi, lo] fc_1;   tail: 1 aux frag, 2 x [hi, lo] fc_out;   then the C table
constexpr int DEC16S_BLK = 12 + 1 + 4 + 4;                                // 21 fragments per block
constexpr int DEC16S_FRAGS = NBLK * DEC16S_BLK + 1 + 4;                   // 110
constexpr size_t DEC16S_BYTES = (DEC16S_FRAGS + 1) * FRAG;                // 111 KiB, resident in LDS per workgroup
// precision f32: per block 12 feature frags (48 MFMAs) + 1 aux frag (2 MFMAs used) + 4 + 4; tail 1 + 4
constexpr int DEC32_FRAGS = NBLK * (12 + 1 + 4 + 4) + 1 + 4;              // 110
constexpr size_t DEC32_BYTES = (DEC32_FRAGS + 1) * FRAG;                  // 111 KiB: fragments + C table chunk (LDS-DMA image)

// w16s: f16x3 split [hi, lo] fragment pairs (2 * nfrag16); wbf: bf16 fragments (f16 fragment layout), derived from w32;
// c32h / c32s / c32b: conv32 images (giga_conv32_geom.h) in f16, f16x3 [hi, lo] pairs (2 * nfragc32) and bf16
// ---- f16-class conv_in (giga_encoder.hip: convin_project_kernel<.., SPLIT = true>): which tap sits in K slot (g, e) of the single
// K = 32 step (lane k-group g = lane >> 4, half e of its 8).  The staged sub-volume of those instantiations has rows of
// CI16_RS = 56 words and slabs of CI16_SLAB = 688 words, and a half-wave holds k-groups (0, 1) or (2, 3): its 32 lanes read, for
// one e, 8 consecutive words in two rows per k-group -- banks b + {0..7, 24..31} (56 = 24 mod 32) -- and the partner k-group must
// sit 16 banks further for the four runs to be disjoint.  An address is dx * SLAB + dy * RS + dz (+ the voxel), SLAB = 16 and
// 2 * RS = 16 mod 32: partners differ by one step in dx (same dy, dz) or are (dy = 0, dy = 2) of one (dx, dz).  Per dz the 3 x 3
// grid of (dx, dy) gives four such pairs and one single tap, whose partner slot has weight ZERO and reads a valid neighbour:
// 12 real pairs + 3 (real, zero) + 1 (zero, zero) = the 16 pairs of the two half-waves.  Every ds_read_b32 of the gather is
// conflict-free (tests/test_abi_and_host.py checks the table: every tap once, the partner rule, 16 cycles per unit in a bank model).
// Tap index = dx * 9 + dy * 3 + dz (the reference's Conv3d weight order, voxels.py:89-97: kernel index [kx][ky][kz] over the
// grid's (x, y, z)).  ci16_tap: the tap whose WEIGHT the slot carries (-1: zero); ci16_read: the tap whose VOXEL the slot reads.
constexpr int CI16_RS = 56, CI16_SLAB = 12 * CI16_RS + 16;
constexpr int ci16_pair_tap(int p, int side) {           // pair p = 5 * dz + q, side 0 / 1
    const int dz = p / 5, q = p % 5;
    if (p >= 15) return -1;
    // (dx, dy) of the two sides
    const int dxa[5] = {0, 2, 0, 0, 2}, dya[5] = {0, 0, 2, 1, 1};
    const int dxb[5] = {1, 2, 1, 1, 1}, dyb[5] = {0, 2, 2, 1, 1};
    if (q == 4 && side == 1) return -1;                  // the single tap's partner: zero weight
    return side == 0 ? dxa[q] * 9 + dya[q] * 3 + dz : dxb[q] * 9 + dyb[q] * 3 + dz;
}
constexpr int ci16_pair_read(int p, int side) {
    const int pp = p >= 15 ? 0 : p;                      // the (zero, zero) pair reads what pair 0 reads
    const int dz = pp / 5, q = pp % 5;
    const int dxa[5] = {0, 2, 0, 0, 2}, dya[5] = {0, 0, 2, 1, 1};
    const int dxb[5] = {1, 2, 1, 1, 1}, dyb[5] = {0, 2, 2, 1, 1};
    return side == 0 ? dxa[q] * 9 + dya[q] * 3 + dz : dxb[q] * 9 + dyb[q] * 3 + dz;
}
constexpr int ci16_tap(int g, int e) { return ci16_pair_tap(8 * (g >> 1) + e, g & 1); }
constexpr int ci16_read(int g, int e) { return ci16_pair_read(8 * (g >> 1) + e, g & 1); }

struct ConvPackOff { size_t w16, w32, bias, w16s, wbf; int nfrag16, nfrag32; size_t c32h, c32s, c32b; int nfragc32;
                     size_t wino; };   // Winograd-domain fp32 image of a 3x3 layer (giga_wino.h), 16 * cin * cout floats; 0 bytes otherwise
struct PackOff {
    size_t convin_w;        // fp32 [2][7][64]  B operands (channel half, K-step of 4 taps; tap 27 = 0)
    size_t convin_b;        // fp32 [32]
    ConvPackOff conv[NCONV];
    size_t dec16[NHEADS], dec32[NHEADS];
    size_t dec16f[NHEADS], dec32f[NHEADS];   // the same heads with the encoder's final 1x1 conv folded into fc_c (see giga_pack.cpp)
    size_t dec16s[NHEADS], dec16sf[NHEADS];  // f16x3 split images (plain, folded)
    size_t convin_ws;       // f16x3 split conv_in B operands: [2 channel halves][hi, lo] fragments of v_mfma_f32_16x16x32_f16
    size_t dect[NHEADS];    // bf16 forward images of the bf16 training decoder (giga_dect.h), derived from dec32
    size_t stamp;           // PackStamp: magic, ABI version, blob size (giga_packed_check)
    size_t total;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline PackOff pack_offsets() {
    PackOff o{};
    size_t at = 0;
    o.convin_w = at; at += 14 * 64 * sizeof(float);
    o.convin_b = at; at += align_up(CD * sizeof(float), 256);
    for (int l = 0; l < NCONV; ++l) {
        const ConvLayerDesc& d = kConv[l];
        const int cin = d.cin0 + d.cin1;
        const int nblk = d.cout / 16 * conv_nsub(d);
        o.conv[l].nfrag16 = nblk * conv_taps(d) * (cin / 32);
        o.conv[l].nfrag32 = nblk * conv_taps(d) * (cin / 16);
        o.conv[l].w16 = at; at += o.conv[l].nfrag16 * FRAG;
        o.conv[l].w32 = at; at += o.conv[l].nfrag32 * FRAG;
        o.conv[l].bias = at; at += align_up(d.cout * sizeof(float), 256);
    }
    for (int h = 0; h < NHEADS; ++h) {
        o.dec16[h] = at; at += align_up(DEC16_BYTES, 256);
        o.dec32[h] = at; at += align_up(DEC32_BYTES, 256);
    }
    for (int h = 0; h < NHEADS; ++h) {
        o.dec16f[h] = at; at += align_up(DEC16_BYTES, 256);
        o.dec32f[h] = at; at += align_up(DEC32_BYTES, 256);
    }
    for (int h = 0; h < NHEADS; ++h) {       // appended: the offsets above are unchanged from ABI version 1 blobs
        o.dec16s[h] = at; at += align_up(DEC16S_BYTES, 256);
        o.dec16sf[h] = at; at += align_up(DEC16S_BYTES, 256);
    }
    for (int l = 0; l < NCONV; ++l) { o.conv[l].w16s = at; at += (size_t)2 * o.conv[l].nfrag16 * FRAG; }
    o.convin_ws = at; at += 4 * FRAG;
    for (int l = 0; l < NCONV; ++l) { o.conv[l].wbf = at; at += (size_t)o.conv[l].nfrag16 * FRAG; }
    // conv32 (round 4): 32x32x16 A-operand fragments, [slice = sub * (cout / 32) + cs][tap][k-chunk of 16 input channels]
    for (int l = 0; l < NCONV; ++l) {
        const ConvLayerDesc& d = kConv[l];
        o.conv[l].nfragc32 = conv_nsub(d) * (d.cout / 32) * conv_taps(d) * ((d.cin0 + d.cin1) / 16);
        o.conv[l].c32h = at; at += (size_t)o.conv[l].nfragc32 * FRAG;
        o.conv[l].c32s = at; at += (size_t)2 * o.conv[l].nfragc32 * FRAG;
        o.conv[l].c32b = at; at += (size_t)o.conv[l].nfragc32 * FRAG;
    }
    for (int h = 0; h < NHEADS; ++h) { o.dect[h] = at; at += align_up(DEC16_BYTES, 256); }     // round 5 (ABI 2)
    for (int l = 0; l < NCONV; ++l) {                                                            // round 6 (ABI 3): giga_wino.h
        o.conv[l].wino = at;
        if (kConv[l].kind == CONV3) at += (size_t)16 * (kConv[l].cin0 + kConv[l].cin1) * kConv[l].cout * sizeof(float);
    }
    o.stamp = at; at += 256;
    o.total = at;
    return o;
}

// ----- backward blob (training, fp32 only) ----------------------------------------------------------
// conv16 fragments of the DATA-GRADIENT convolution of every U-Net layer:
//   CONV3  W'[ci][co][tap'] = W[co][ci][8 - tap']   (180-degree flip, channels swapped)
//   CONV1  W'[ci][co]       = W[co][ci]
//   UPCONV W'[ci][co][d]    = W[ci][co][d]          consumed by the DOWN kind (4 taps = the 2x2 sub-pixels)
// followed by the transposed decoder matrices (see giga_decoder_bwd.hip).
struct BwdPackOff {
    size_t conv[NCONV];          // fragment offset of layer l's dgrad image
    int nfrag[NCONV];
    size_t dec[NHEADS];          // transposed decoder matrices of head h
    size_t convbf[NCONV];        // bf16 images of the dgrad fragments (f16 fragment layout, nfrag[l] / 2 fragments)
    size_t dect[NHEADS];         // bf16 transposed decoder matrices of the bf16 training decoder (giga_dect.h), derived from dec
    size_t wino[NCONV];          // Winograd images of the 3x3 layers' DATA-GRADIENT convolutions (giga_wino.h; cin' = cout, cout' = cin, taps flipped)
    size_t stamp;                // PackStamp
    size_t total;
};
// decoder backward image per head: 5 blocks x (Wc^T: 3 row blocks x 4 frags, W0^T 4 frags, W1^T 4 frags)
// + Wout (4 x 32 floats, plain) = 100 fragments + 512 B
constexpr int DECB_FRAGS = NBLK * (12 + 4 + 4);
constexpr size_t DECB_BYTES = (DECB_FRAGS + 1) * FRAG;

inline BwdPackOff bwd_pack_offsets() {
    BwdPackOff o{};
    size_t at = 0;
    for (int l = 0; l < NCONV; ++l) {
        const ConvLayerDesc& d = kConv[l];
        const int cin = d.cin0 + d.cin1;                    // forward channels in / out
        const int taps = d.kind == CONV3 ? 9 : d.kind == UPCONV ? 4 : 1;
        o.nfrag[l] = (cin / 16) * taps * (d.cout / 16);     // (cout'/16) x taps x (cin'/16), primes = swapped
        o.conv[l] = at; at += (size_t)o.nfrag[l] * FRAG;
    }
    for (int h = 0; h < NHEADS; ++h) { o.dec[h] = at; at += DECB_BYTES; }
    for (int l = 0; l < NCONV; ++l) { o.convbf[l] = at; at += (size_t)(o.nfrag[l] / 2) * FRAG; }
    for (int h = 0; h < NHEADS; ++h) { o.dect[h] = at; at += (size_t)(NBLK * 10 + 1) * FRAG; }  // DECT_BWD_BYTES (giga_dect.h)
    for (int l = 0; l < NCONV; ++l) {                                                            // round 6 (ABI 3)
        o.wino[l] = at;
        if (kConv[l].kind == CONV3) at += (size_t)16 * (kConv[l].cin0 + kConv[l].cin1) * kConv[l].cout * sizeof(float);
    }
    o.stamp = at; at += 256;
    o.total = at;
    return o;
}

// Last 256 bytes of both blobs: which layout the blob was packed for.  The layout changes with the ABI version (new images are appended,
// slot orders change); a blob of another version run through this library reads weights from the wrong places, silently.
// giga_packed_check() validates a HOST copy before it is uploaded; the kernels cannot afford to.
struct PackStamp { char magic[8]; int abi_version; int backward; unsigned long long total; };
constexpr int PACK_ABI_VERSION = 3;               // == GIGA_ABI_VERSION (include/giga_hip.h; giga_capi.hip static_asserts it)

// feature index held in D-register r of lane-half hi after a 32x32 MFMA with weights as the A
// operand (rows = output features):  row = (r&3) + 8*(r>>2) + 4*hi   (C/D map, dtype independent)
inline int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

}  // namespace giga
