// conv32: geometry of the weight-stationary, LDS-image-resident convolution of the f16-class U-Net (giga_conv32.h).
//
// Everything here is plain integer arithmetic shared by the gfx950 kernel, the host-side weight packer and the CPU emulation
// of the kernel's data movement (tests/emu/conv32_emu.cpp): which LDS byte a staged value lands on, which LDS bytes a lane
// reads for (tile, tap, k-chunk), which output pixel a lane's accumulators belong to, which packed fragment a wave holds.
//
// Reference layers: UNet.forward, ConvONets/encoder/unet.py:225-239 (conv3x3 :14-23, upconv2x2 :25-31, conv1x1 :39-45,
// MaxPool2d :64, concat order (from_up, from_down) :109).
//
// The shape of the computation (one U-Net layer, one GROUP of 8 workgroups on one XCD, G plane images):
//   * The G images of the group are stacked into one tall image.  For the 3x3 layers two neighbouring images share ONE zero row
//     (the bottom padding of one is the top padding of the next): stacked row s = g * (H + 1) + 1 + y, rows g * (H + 1) are
//     zero.  Member m of the group computes the stacked output rows [sA, sB) of member_rows(): a balanced eighth.
//   * A member walks its rows in SUB-BANDS of at most RBMAX rows: the haloed sub-band -- rows sb-1 .. sb+R, P = W + 2 pixels per
//     row with the zero columns in place -- is staged ONCE into LDS, pixel-major, PS bytes per pixel (all input channels of the
//     pixel + 16 bytes of padding: PS / 16 is odd, so 16 consecutive pixels hit 16 different 16-byte bank groups and every
//     ds_read_b128 below is conflict-free).
//   * Output positions of a sub-band are numbered linearly, o = orow * P + ocol (the two pad columns of every row produce
//     values that are discarded: 5 % at 40 x 40, 9 % at 20 x 20).  A TILE is 32 consecutive o.  The input pixel of output o for
//     tap (ky, kx) is buffer pixel o + ky * P + kx: a CONSTANT offset, so the B operand of (tile, tap, 16-channel chunk) is one
//     ds_read_b128 per lane at `lane base + immediate`.
//   * v_mfma_f32_32x32x16_{f16,bf16}: A operand = weights (rows = 32 output channels of a SLICE), B operand = 32 pixels of the
//     tile.  A wave keeps the fragments of its slice(s) for all taps and input channels IN REGISTERS for the whole layer
//     (9 taps x Cin/16 chunks x 4 VGPRs per slice: 72 ... 288 VGPRs; one wave per SIMD, 512 VGPRs), so the only LDS traffic of
//     the MFMA loop is the B operand: 1 KiB per MFMA (SPW = 1) or per two MFMAs (SPW = 2) -- 50 / 25 % of the LDS read rate,
//     against 1.5 KiB per 16-clock MFMA for conv16.
//   * Row -> output channel map of a slice: D register r of lane half hi holds row (r&3) + 8 (r>>2) + 4 hi; the packer puts
//     output channel 16 hi + r there (c32_row_cout), so a lane's 16 accumulators are 16 CONSECUTIVE output channels of one
//     pixel: 32 bytes of f16 (64 of fp32) per lane and store.
#pragma once
#include "giga_layout.h"

#if defined(__HIPCC__)
#define GIGA_HD __host__ __device__ inline
#else
#define GIGA_HD inline
#endif

namespace giga {

constexpr int C32_NATIVE = 0;     // f16 activations in memory, f16 MFMA                                   (precision 1)
constexpr int C32_SPLIT = 1;      // fp32 activations in memory, f16x3 split operands (hi, lo), three f16 MFMAs (precision 2)
constexpr int C32_BF16 = 2;       // fp32 activations in memory, bf16 operands, one bf16 MFMA                 (precision 3)

constexpr int C32_NW = 4;                          // waves per workgroup: one per SIMD, up to 512 VGPRs each
constexpr int C32_GROUP = 8;                       // workgroups per group (unet_mega_kernel's groups)
constexpr int C32_LDS = 160 * 1024 - 1024;         // dynamic LDS of the kernels (the persistent kernel keeps a few static words)
constexpr int C32_TAIL = 32;                       // pixels a tile may read past the staged sub-band (discarded lanes only)

// output channel (inside its 32-channel slice) held by A-operand row i
constexpr int c32_row_cout(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

template <int MODE_, int KIND_, int C0_, int C1_, int COUT_, int H_, int W_, bool POOLIN_, int SPW_, int KPARTS_ = 1>
struct C32 {
    static constexpr int MODE = MODE_, KIND = KIND_, C0 = C0_, C1 = C1_, COUT = COUT_, H = H_, W = W_;
    static constexpr bool POOLIN = POOLIN_;        // the input is max_pool2d(source, 2, 2): the source is 2H x 2W and is pooled while it is staged
    static constexpr int SPW = SPW_;               // weight slices a wave holds (and MFMAs per B-operand read)
    static constexpr int KPARTS = KPARTS_;         // the input channels are walked in this many parts (weights reloaded per part)
    static constexpr int HALO = KIND == CONV3 ? 1 : 0;
    static constexpr int P = W + 2 * HALO;         // LDS row pitch in pixels
    static constexpr int SR = H + HALO;            // stacked rows per image
    static constexpr int CIN = C0 + C1;
    static constexpr int TAPS = KIND == CONV3 ? 9 : 1;
    static constexpr int NSUB = KIND == UPCONV ? 4 : 1;
    static constexpr int EB = MODE == C32_SPLIT ? 4 : 2;      // LDS bytes per channel (split: hi + lo)
    static constexpr int ES = MODE == C32_NATIVE ? 2 : 4;     // bytes per element in memory
    static constexpr int PS = CIN * EB + 16;       // LDS bytes per pixel
    static constexpr int IPP = CIN / 8;            // staging items (8 channels) per pixel
    static constexpr int ILB = 8 * EB;             // LDS bytes per item: native / bf16 16, split 32 ([8 hi | 8 lo])
    static constexpr int KC = CIN / 16;            // k-chunks (16 channels = one MFMA)
    static constexpr int KCP = KC / KPARTS;        // k-chunks per part
    static constexpr int KB = 16 * EB;             // LDS bytes per k-chunk of a pixel
    static constexpr int HB = 8 * EB;              // byte offset of lane half hi = 1 (channels 8..15 of the chunk)
    static constexpr int CS = COUT / 32;           // 32-channel slices per sub-output
    static constexpr int NS = NSUB * CS;           // weight slices of the layer
    static constexpr int SG = NS / SPW;            // slice groups: the waves of a workgroup are dealt out over them
    static constexpr int TL = C32_NW / SG;         // waves per slice group = tile lanes
    static constexpr int NTB = SPW == 1 ? 2 : 1;   // tiles a wave has in flight (two independent accumulator chains at least)
    static constexpr int NOP = MODE == C32_SPLIT ? 2 : 1;     // operand registers per fragment ([hi, lo])
    static constexpr int RBMAX = (C32_LDS / PS - C32_TAIL) / P - 2 * HALO;     // rows per sub-band
    static constexpr int IH = POOLIN ? 2 * H : H, IW = POOLIN ? 2 * W : W;     // source grid
    static constexpr int OH = KIND == UPCONV ? 2 * H : H, OW = KIND == UPCONV ? 2 * W : W;
    static constexpr int NFRAG = NS * TAPS * KC;   // fragments (x NOP) of the layer in the packed blob
    static_assert(CIN % 16 == 0 && COUT % 32 == 0, "channel counts");
    static_assert(NS % SPW == 0 && (SG == 1 || SG == 2 || SG == 4), "slice groups must divide the four waves");
    static_assert(KC % KPARTS == 0, "k parts");
    static_assert((PS / 16) % 2 == 1, "odd pixel stride in 16-byte units: conflict-free ds_read_b128");
    static_assert(RBMAX >= 1, "one row must fit");
    static_assert(!(POOLIN && C1 > 0), "pooled inputs are single tensors");

    // stacked output rows [sA, sB) of member m (0..7) of a group with G images
    static GIGA_HD void member_rows(int m, int G, int& sA, int& sB) {
        const int NR = G * SR - HALO;
        sA = HALO + (m * NR) / C32_GROUP;
        sB = HALO + ((m + 1) * NR) / C32_GROUP;
    }
    // sub-bands of a member: n bands of `rows` rows (the last one shorter)
    static GIGA_HD void sub_bands(int sA, int sB, int& n, int& rows) {
        const int r = sB - sA;
        n = (r + RBMAX - 1) / RBMAX;
        rows = n ? (r + n - 1) / n : 0;
    }
    static GIGA_HD int n_tiles(int R) { return (R * P - 2 * HALO + 31) / 32; }
    static GIGA_HD int lds_bytes(int R) { return ((R + 2 * HALO) * P + C32_TAIL) * PS; }

    // ---- staging -------------------------------------------------------------------------------------------------------------
    // The REAL input rows of the group (rows inside an image) are numbered rr = g * H + y; an ITEM is 8 channels of one pixel,
    // RI items per real row.  The real rows a sub-band needs are a contiguous range [rrA, rrB) and their items are contiguous
    // in memory (a single dense tensor: byte offset = item index * item bytes), so thread t takes items t, t + 256, ... : the
    // source address is linear, and the (row, position) decomposition that the LDS address needs is carried along
    // incrementally (Cur: two adds and two compares per item, no division).  Zero rows and the two zero columns of every row
    // are written separately (pad_pixel).
    static constexpr int RI = W * IPP;
    static constexpr int NTHR = C32_NW * 64;
    // number of real rows whose stacked index is < s
    static GIGA_HD int real_rows_below(int s, int G) {
        if (HALO == 0) return s < 0 ? 0 : (s > G * H ? G * H : s);
        if (s <= 0) return 0;
        if (s >= G * SR) return G * H;
        const int g = s / SR, r = s - g * SR;
        return g * H + (r > 0 ? r - 1 : 0);
    }
    static GIGA_HD void real_rows(int sb, int R, int G, int& rrA, int& rrB) {
        rrA = real_rows_below(sb - HALO, G);
        rrB = real_rows_below(sb + R + HALO, G);
    }
    struct Cur { int rr, c, g, y; };                     // real row, item inside the row, image, row inside the image
    static GIGA_HD Cur cur_init(int rrA, int j) {
        Cur k;
        const int q = j / RI;
        k.rr = rrA + q; k.c = j - q * RI;
        k.g = k.rr / H; k.y = k.rr - k.g * H;
        return k;
    }
    static GIGA_HD void cur_next(Cur& k) {               // j += NTHR
        constexpr int DR = NTHR / RI, DC = NTHR % RI;
        static_assert(DR + 1 < H, "at most one image boundary per step");
        int d = DR;
        k.c += DC;
        if (k.c >= RI) { k.c -= RI; ++d; }
        k.rr += d; k.y += d;
        if (k.y >= H) { k.y -= H; ++k.g; }
    }
    static GIGA_HD int cur_x(const Cur& k) { return k.c / IPP; }
    static GIGA_HD int cur_ch(const Cur& k) { return 8 * (k.c % IPP); }
    // LDS byte offset of the item in the sub-band that starts at stacked row sb
    static GIGA_HD int cur_lds(const Cur& k, int sb) {
        const int brow = k.g * SR + HALO + k.y - (sb - HALO);
        return (brow * P + cur_x(k) + HALO) * PS + (k.c % IPP) * ILB;
    }
    // is row `s` (stacked) one this sub-band computes, i.e. not a halo row?  (the pooled write-through of the POOLIN layers)
    static GIGA_HD bool cur_own(const Cur& k, int sb, int R) {
        const int s = k.g * SR + HALO + k.y;
        return s >= sb && s < sb + R;
    }
    // source pixel (in pixels of the IH x IW source grid of the group) of sub-position q (POOLIN: the 2x2 window; else q = 0)
    static GIGA_HD int cur_src_pixel(const Cur& k, int q) {
        return POOLIN ? (2 * k.rr + (q >> 1)) * IW + 2 * cur_x(k) + (q & 1) : k.rr * W + cur_x(k);
    }
    // buffer pixel q of the sub-band (0 .. (R + 2 HALO) * P): must it be written as zeros?
    static GIGA_HD int n_buf_pixels(int R) { return (R + 2 * HALO) * P; }
    static GIGA_HD bool pad_pixel(int q, int sb, int G) {
        if (HALO == 0) return false;
        const int brow = q / P, col = q - brow * P;
        const int s = sb - HALO + brow;
        const int g = s / SR, r = s - g * SR;
        return r == 0 || g >= G || col == 0 || col == P - 1;
    }
    // LDS byte offset a lane adds to (tile, tap, chunk) offsets
    static GIGA_HD int lane_base(int lane) { return (lane & 31) * PS + (lane >> 5) * HB; }
    static constexpr int tile_step() { return 32 * PS; }
    static constexpr int tap_off(int tap) { return KIND == CONV3 ? ((tap / 3) * P + tap % 3) * PS : 0; }
    static constexpr int kc_off(int kc) { return kc * KB; }      // (split: the lo half of the operand sits 16 bytes further)

    // output pixel of lane column n of tile t
    struct Out { int g, y, x; bool valid; };
    static GIGA_HD Out out_pixel(int t, int n, int sb, int R) {
        Out o;
        const int lin = 32 * t + n;
        const int orow = lin / P;
        o.x = lin - orow * P;
        const int s = sb + orow;
        o.g = s / SR;
        o.y = s - o.g * SR - HALO;
        o.valid = orow < R && o.x < W && o.y >= 0;
        return o;
    }
    // index (in pixels, inside the group's output range) of the pixel that sub-output `sub` of (g, y, x) is stored to
    static GIGA_HD int out_index(int g, int y, int x, int sub) {
        return KIND == UPCONV ? (g * OH + 2 * y + (sub >> 1)) * OW + 2 * x + (sub & 1) : (g * H + y) * W + x;
    }
    // packed fragment of (slice = sub * CS + cs, tap, k-chunk); split blobs hold the pair (2 f, 2 f + 1) = (hi, lo)
    static constexpr int frag(int slice, int tap, int kc) { return (slice * TAPS + tap) * KC + kc; }
};

// The U-Net on conv32.  X(layer, KIND, C0, C1, COUT, H, W, POOLIN, SPW f16 / bf16, SPW f16x3): layer order of giga_layout.h::kConv;
// SPW = weight slices (32 output channels) per wave = MFMAs per LDS operand read, bounded by the register file (a slice is
// TAPS * CIN / 16 fragments of 4 VGPRs, twice that in the split mode).  Layers 2 and 4 read the un-pooled skip tensors and
// pool them while staging (and write the pooled tensors Q0 / Q1, which the workspace layout of the C ABI exposes).
#define GIGA_UNET32_LAYERS(X)                            \
    X(0, CONV3, 32, 0, 32, 40, 40, false, 1, 1)          \
    X(1, CONV3, 32, 0, 32, 40, 40, false, 1, 1)          \
    X(2, CONV3, 32, 0, 64, 20, 20, true, 2, 2)           \
    X(3, CONV3, 64, 0, 64, 20, 20, false, 2, 1)          \
    X(4, CONV3, 64, 0, 128, 10, 10, true, 2, 1)          \
    X(5, CONV3, 128, 0, 128, 10, 10, false, 1, 1)        \
    X(6, UPCONV, 128, 0, 64, 10, 10, false, 4, 4)        \
    X(7, CONV3, 64, 64, 64, 20, 20, false, 1, 1)         \
    X(8, CONV3, 64, 0, 64, 20, 20, false, 2, 1)          \
    X(9, UPCONV, 64, 0, 32, 20, 20, false, 4, 4)         \
    X(10, CONV3, 32, 32, 32, 40, 40, false, 1, 1)        \
    X(11, CONV3, 32, 0, 32, 40, 40, false, 1, 1)         \
    X(12, CONV1, 32, 0, 32, 40, 40, false, 1, 1)
template <int MODE, int L> struct U32Layer;
#define X(l, KIND, C0, C1, COUT, H, W, POOLIN, SN, SS)                                                            \
    template <int MODE> struct U32Layer<MODE, l> {                                                                \
        using G = C32<MODE, KIND, C0, C1, COUT, H, W, POOLIN, (MODE == C32_SPLIT ? SS : SN)>;                     \
        static constexpr bool RELU = KIND == CONV3;                                                               \
    };
GIGA_UNET32_LAYERS(X)
#undef X

}  // namespace giga
