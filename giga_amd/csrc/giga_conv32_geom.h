// conv32: geometry of the LDS-image-resident 32x32x16 convolution of the f16-class U-Net (giga_conv32.h).
//
// Everything here is plain integer arithmetic shared by the gfx950 kernel, the host-side weight packer and the CPU emulation
// of the kernel's data movement (tests/emu/conv32_emu.cpp): which LDS byte a staged value lands on, which LDS bytes a lane
// reads for (tile, tap, k-chunk), which output pixel a lane's accumulators belong to, which packed fragment sits where.
//
// Reference layers: UNet.forward, ConvONets/encoder/unet.py:225-239 (conv3x3 :14-23, upconv2x2 :25-31, conv1x1 :39-45,
// MaxPool2d :64, concat order (from_up, from_down) :109).
//
// The shape of the computation (one U-Net layer, one GROUP of 8 workgroups on one XCD, G plane images):
//   * The G images of the group are stacked into one tall image.  For the 3x3 layers two neighbouring images share ONE zero row
//     (the bottom padding of one is the top padding of the next): stacked row s = g * (H + 1) + 1 + y, rows g * (H + 1) are
//     zero.  The 8 members of the group are SGM slice groups x 8 / SGM row bands: member m computes the output channels of
//     slice group m % SGM for the stacked rows [sA, sB) of band m / SGM (member_rows(): balanced).  SGM > 1 where a layer's
//     weights are too large to sit in every workgroup's LDS.
//   * The member's weight fragments (its SPM slices of 32 output channels, all taps, all input channels) are copied ONCE into
//     LDS by LDS-DMA -- one copy per workgroup, shared by its eight waves -- and stay there for the layer.
//   * A member walks its rows in SUB-BANDS of at most RBMAX rows: the haloed sub-band -- rows sb-1 .. sb+R, P = W + 2 pixels per
//     row with the zero columns in place -- is staged ONCE into LDS behind the weights, pixel-major, PS bytes per pixel (all
//     input channels of the pixel + 16 bytes of padding: PS / 16 is odd, so 16 consecutive pixels hit 16 different 16-byte
//     bank groups and every ds_read_b128 below is conflict-free).
//   * Output positions of a sub-band are numbered linearly, o = orow * P + ocol (the two pad columns of every row produce
//     values that are discarded: 5 % at 40 x 40, 9 % at 20 x 20).  A TILE is 32 consecutive o.  The input pixel of output o for
//     tap (ky, kx) is buffer pixel o + ky * P + kx: a CONSTANT offset, so the B operand of (tile, tap, 16-channel chunk) is one
//     ds_read_b128 per lane at `lane base + immediate`.
//   * v_mfma_f32_32x32x16_{f16,bf16}: A operand = weights (rows = 32 output channels of a slice; a fragment is 1 KiB,
//     lane-linear: one conflict-free ds_read_b128), B operand = 32 pixels of a tile.  A wave works on a register tile of
//     NTB tiles x SPW slices: per (tap, k-chunk) it reads SPW A fragments and NTB B fragments and issues NTB * SPW MFMAs, i.e.
//     (NTB + SPW) / (NTB * SPW) KiB of LDS reads per MFMA: 1 KiB (2 x 2) = half the LDS read rate, against 1.5 KiB per
//     16-clock MFMA (three times the rate) for conv16's 16x16x32 tiles fed from wave-private patches.
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

constexpr int C32_NW = 8;                          // waves per workgroup: two per SIMD, up to 256 VGPRs each
constexpr int C32_GROUP = 8;                       // workgroups per group (unet_mega_kernel's groups)
constexpr int C32_LDS = 160 * 1024 - 1024;         // dynamic LDS of the kernels (the persistent kernel keeps a few static words)
constexpr int C32_BIAS_BYTES = 512;                // behind the C32_LDS bytes of weights + images: the layer's biases (two sets of 64 for a fused pair)
constexpr int C32_LDS_TOTAL = C32_LDS + C32_BIAS_BYTES;
constexpr int C32_TAIL = 32;                       // pixels a tile may read past the staged sub-band (discarded lanes only)
constexpr int C32_NTB_MAX = 3;                     // tiles of a wave's register tile at most

// output channel (inside its 32-channel slice) held by A-operand row i
constexpr int c32_row_cout(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

template <int MODE_, int KIND_, int C0_, int C1_, int COUT_, int H_, int W_, bool POOLIN_, int SPW_, int SGM_, int KP_ = 1>
struct C32 {
    static constexpr int MODE = MODE_, KIND = KIND_, C0 = C0_, C1 = C1_, COUT = COUT_, H = H_, W = W_;
    static constexpr bool POOLIN = POOLIN_;        // the input is max_pool2d(source, 2, 2): the source is 2H x 2W and is pooled while it is staged
    static constexpr int SGM = SGM_;               // slice groups among the 8 members of a group
    static constexpr int KP = KP_;                 // channel PARTS: the input channels are walked in KP passes, each with its own weight
                                                   // fill and its own staged image (CIN / KP channels per pixel); accumulators persist
    static constexpr int HALO = KIND == CONV3 ? 1 : 0;
    static constexpr int P = W + 2 * HALO;         // LDS row pitch in pixels
    static constexpr int SR = H + HALO;            // stacked rows per image
    static constexpr int CIN = C0 + C1;
    static constexpr int TAPS = KIND == CONV3 ? 9 : 1;
    static constexpr int NSUB = KIND == UPCONV ? 4 : 1;
    static constexpr int EB = MODE == C32_SPLIT ? 4 : 2;      // LDS bytes per channel (split: hi + lo)
    static constexpr int ES = MODE == C32_NATIVE ? 2 : 4;     // bytes per element in memory
    static constexpr int CINP = CIN / KP;          // channels per part
    static constexpr int PS = CINP * EB + 16;      // LDS bytes per pixel
    static constexpr int IPP = CINP / 8;           // staging items (8 channels) per pixel
    static constexpr int ILB = 8 * EB;             // LDS bytes per item: native / bf16 16, split 32 ([8 hi | 8 lo])
    static constexpr int KC = CIN / 16;            // k-chunks (16 channels = one MFMA) of the layer
    static constexpr int KCP = CINP / 16;          // ... of a part
    static constexpr int KB = 16 * EB;             // LDS bytes per k-chunk of a pixel
    static constexpr int HB = 8 * EB;              // byte offset of lane half hi = 1 (channels 8..15 of the chunk)
    static constexpr int CS = COUT / 32;           // 32-channel slices per sub-output
    static constexpr int NS = NSUB * CS;           // weight slices of the layer
    static constexpr int SPM = NS / SGM;           // slices per member
    static constexpr int SPW = SPW_ < SPM ? SPW_ : SPM;       // slices of a wave's register tile (A fragments read per step)
    // the same layer with one-slice register tiles: same LDS contents (weights, image), twice the tile tasks -- for sub-bands so
    // small that the SPW-slice tasks would leave most waves idle (the 20 x 20 and 10 x 10 layers of one or two scenes)
    using Thin = C32<MODE, KIND, C0, C1, COUT, H, W, POOLIN, 1, SGM_, KP_>;
    static constexpr int NSP = SPM / SPW;          // slice passes of a tile batch
    static constexpr int NBANDS = C32_GROUP / SGM; // row bands of a group
    static constexpr int NOP = MODE == C32_SPLIT ? 2 : 1;     // operand registers per fragment ([hi, lo])
    static constexpr int NIT = TAPS * KCP;         // MFMA steps of a (tile, slice, part)
    static constexpr int WFR = SPM * NIT * NOP;    // 1-KiB fragments resident in LDS (one part)
    static constexpr int WBYTES = WFR * 1024;      // ... the image region starts here
    static constexpr int NTBM = SPW == 1 ? C32_NTB_MAX : 2;   // tiles of a register tile at most (NTB * SPW <= 4 accumulators)
    static constexpr int RB_LDS = ((C32_LDS - WBYTES) / PS - C32_TAIL) / P - 2 * HALO;
    static constexpr int RB_ONE = (32 * C32_NW * NTBM + 2 * HALO) / P;        // rows that are at most one batch per wave
    static constexpr int RBMAX = KP > 1 && RB_ONE < RB_LDS ? RB_ONE : RB_LDS; // rows per sub-band (KP > 1: accumulators live across the parts)
    static constexpr int IH = POOLIN ? 2 * H : H, IW = POOLIN ? 2 * W : W;     // source grid
    static constexpr int OH = KIND == UPCONV ? 2 * H : H, OW = KIND == UPCONV ? 2 * W : W;
    static constexpr int NFRAG = NS * TAPS * KC;   // fragments (x NOP) of the layer in the packed blob
    static constexpr int NTHR = C32_NW * 64;
    static_assert(CIN % 16 == 0 && COUT % 32 == 0, "channel counts");
    static_assert(NS % SGM == 0 && SPM % SPW == 0 && C32_GROUP % SGM == 0, "slice grouping");
    static_assert((PS / 16) % 2 == 1, "odd pixel stride in 16-byte units: conflict-free ds_read_b128");
    static_assert(RBMAX >= 1, "the weights and one row must fit");
    static_assert(!(POOLIN && C1 > 0), "pooled inputs are single tensors");
    static_assert(CIN % KP == 0 && CINP % 16 == 0, "channel parts");
    static_assert(KP == 1 || SPM == SPW, "channel parts keep ONE register tile per wave alive across the parts");
    static_assert(KP == 1 || C1 == 0 || (KP == 2 && C0 == C1), "parts of a concatenated input are its two tensors");

    // member m (0..7) of a group with G images: its slice group and its stacked output rows [sA, sB)
    static GIGA_HD int member_sgm(int m) { return m % SGM; }
    static GIGA_HD void member_rows(int m, int G, int& sA, int& sB) {
        const int NR = G * SR - HALO, band = m / SGM;
        sA = HALO + (band * NR) / NBANDS;
        sB = HALO + ((band + 1) * NR) / NBANDS;
    }
    // sub-bands of a member: n bands of `rows` rows (the last one shorter)
    static GIGA_HD void sub_bands(int sA, int sB, int& n, int& rows) {
        const int r = sB - sA;
        n = (r + RBMAX - 1) / RBMAX;
        rows = n ? (r + n - 1) / n : 0;
    }
    static GIGA_HD int n_tiles(int R) { return (R * P - 2 * HALO + 31) / 32; }
    static GIGA_HD int lds_bytes(int R) { return WBYTES + ((R + 2 * HALO) * P + C32_TAIL) * PS; }
    // tiles of a wave's register tile for a sub-band of NT tiles: as few as keep all eight waves to one batch each
    static GIGA_HD int batch_tiles(int NT) {
        const int n = (NT + C32_NW - 1) / C32_NW;
        return n < 1 ? 1 : (n > NTBM ? NTBM : n);
    }

    // ---- weights ---------------------------------------------------------------------------------------------------------------
    // packed fragment of (slice = sub * CS + cs, tap, k-chunk); split blobs hold the pair (2 f, 2 f + 1) = (hi, lo)
    static constexpr int frag(int slice, int tap, int kc) { return (slice * TAPS + tap) * KC + kc; }
    // LDS fragment c (0 .. WFR) of part `part` of slice group sgm: c = ((sl * TAPS + tap) * KCP + kcp) * NOP + o holds blob
    // fragment (frag(sgm * SPM + sl, tap, part * KCP + kcp)) * NOP + o   (KP == 1: the member's fragments in blob order)
    static GIGA_HD int fill_src(int c, int sgm, int part) {
        const int o = c % NOP, f = c / NOP;
        const int kcp = f % KCP, tap = (f / KCP) % TAPS, sl = f / (KCP * TAPS);
        return frag(sgm * SPM + sl, tap, part * KCP + kcp) * NOP + o;
    }
    // LDS byte offset of the A operand of (member slice sl, step it = tap * KCP + kcp, operand o)
    static constexpr int w_off(int sl, int it, int o) { return ((sl * NIT + it) * NOP + o) * 1024; }
    // source of part `part`: tensor (0: in0, 1: in1) and first channel inside its pixels, channels per source pixel
    static GIGA_HD int part_tensor(int part) { return KP > 1 && C1 > 0 ? part : 0; }
    static GIGA_HD int part_ch0(int part) { return KP > 1 && C1 == 0 ? part * CINP : 0; }

    // ---- staging -------------------------------------------------------------------------------------------------------------
    // The REAL input rows of the group (rows inside an image) are numbered rr = g * H + y; an ITEM is 8 channels of one pixel,
    // RI items per real row.  The real rows a sub-band needs are a contiguous range [rrA, rrB) and their items are contiguous
    // in memory, so thread t takes items t, t + NTHR, ...  Zero rows and the two zero columns of every row are written separately
    // (pad_pixel).
    static constexpr int RI = W * IPP;
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
    // A thread's position: real row rr, pixel x inside it, row y inside its image, pixel index spix = rr * W + x inside the
    // group's real rows, LDS byte offset of the item.  The channel group v = t % IPP of a thread never changes (IPP divides NTHR
    // and RI), so a step of NTHR items is a step of NTHR / IPP pixels: everything advances by adds and two compares.
    struct Cur { int rr, x, y, spix, lds; };
    static constexpr int ROWB = P * PS;                  // LDS bytes per buffer row
    static constexpr int DXT = NTHR / IPP;               // pixels per step
    static GIGA_HD int thr_ch(int t) { return 8 * (t % IPP); }
    static GIGA_HD Cur cur_init(int rrA, int t, int sb) {
        Cur k;
        const int q = t / RI, c = t - q * RI;
        k.rr = rrA + q; k.x = c / IPP;
        const int g = k.rr / H;
        k.y = k.rr - g * H;
        k.spix = k.rr * W + k.x;
        const int brow = g * SR + HALO + k.y - (sb - HALO);
        k.lds = WBYTES + (brow * P + k.x + HALO) * PS + (t % IPP) * ILB;
        return k;
    }
    static GIGA_HD void cur_next(Cur& k) {               // item index += NTHR
        constexpr int DR = DXT / W, DX = DXT % W;
        static_assert(DR + 1 < H, "at most one image boundary per step");
        int d = DR;
        k.x += DX; k.spix += DXT; k.lds += DR * ROWB + DX * PS;
        if (k.x >= W) { k.x -= W; ++d; k.lds += ROWB - W * PS; }
        k.rr += d; k.y += d;
        if (k.y >= H) { k.y -= H; k.lds += HALO * ROWB; }      // into the next image: over the shared zero row
    }
    // is the item's row one this sub-band computes, i.e. not a halo row?  (the pooled write-through of the POOLIN layers)
    static GIGA_HD bool cur_own(const Cur& k, int sb, int R) {
        const int g = k.rr / H;
        const int s = g * SR + HALO + (k.rr - g * H);
        return s >= sb && s < sb + R;
    }
    // source pixel (in pixels of the IH x IW source grid of the group) of sub-position q (POOLIN: the 2x2 window; else q = 0)
    static GIGA_HD int cur_src_pixel(const Cur& k, int q) {
        return POOLIN ? 2 * (k.spix + k.rr * W) + (q >> 1) * IW + (q & 1) : k.spix;
    }
    // buffer pixel q of the sub-band (0 .. (R + 2 HALO) * P): must it be written as zeros?  (LDS bytes from WBYTES + q * PS)
    static GIGA_HD int n_buf_pixels(int R) { return (R + 2 * HALO) * P; }
    static GIGA_HD bool pad_pixel(int q, int sb, int G) {
        if (HALO == 0) return false;
        const int brow = q / P, col = q - brow * P;
        const int s = sb - HALO + brow;
        if (s < 0) return true;                          // (above the first zero row: the upper halo of a fused pair's first member)
        const int g = s / SR, r = s - g * SR;
        return r == 0 || g >= G || col == 0 || col == P - 1;
    }

    // ---- tiles ---------------------------------------------------------------------------------------------------------------
    // LDS byte offset of a lane's B operand of (tile t, tap, chunk): lane_base + t * tile_step + tap_off + kc_off (+ 16: lo half)
    static GIGA_HD int lane_base(int lane) { return WBYTES + (lane & 31) * PS + (lane >> 5) * HB; }
    static constexpr int tile_step() { return 32 * PS; }
    static constexpr int tap_off(int tap) { return KIND == CONV3 ? ((tap / 3) * P + tap % 3) * PS : 0; }
    static constexpr int kc_off(int kc) { return kc * KB; }

    // output pixel of lane column n of tile t
    struct Out { int g, y, x, orow; bool valid; };
    static GIGA_HD Out out_pixel(int t, int n, int sb, int R) {
        Out o;
        const int lin = 32 * t + n;
        const int orow = lin / P;
        o.orow = orow;
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
};

// ---- FUSED PAIRS --------------------------------------------------------------------------------------------------------------------
// Two consecutive 3x3 layers A -> B of the same resolution whose weights both fit (SGM = 1, KP = 1): a member computes A for its
// band PLUS one row above and below (A's input staged with a halo of two), writes those rows into LDS as B's input image -- zero
// rows and columns in place, in B's pixel format -- and computes B from there.  No group barrier, no store acknowledgement, no
// reload between the two layers: one layer boundary less, paid with (R + 2) / R of A's MFMAs.  A's own rows still go to memory
// (the workspace layout of the C ABI exposes them); the two recomputed rows do not (the neighbours write theirs).
// LDS: [weights A | weights B | image A: (R + 4) rows | image B ("mid"): (R + 2) rows].
template <class GA, class GB>
struct C32Pair {
    static_assert(GA::KIND == CONV3 && GB::KIND == CONV3 && GA::H == GB::H && GA::W == GB::W, "same-resolution 3x3 layers");
    static_assert(GA::COUT == GB::CIN && GB::C1 == 0 && !GB::POOLIN, "B reads A's output");
    static_assert(GA::SGM == 1 && GB::SGM == 1 && GA::KP == 1 && GB::KP == 1 && GA::MODE == GB::MODE, "whole pixels, whole weights");
    static constexpr int WA = GA::WBYTES, WB = GB::WBYTES;
    static constexpr int FIX = 4 * GA::ROWB + 2 * GB::ROWB + C32_TAIL * (GA::PS + GB::PS);
    static constexpr int RBMAX = (C32_LDS - WA - WB - FIX) / (GA::ROWB + GB::ROWB);      // rows of B per sub-band
    static_assert(RBMAX >= 1, "a fused pair must fit at least one row");
    static GIGA_HD int imgA_bytes(int R) { return ((R + 4) * GA::P + C32_TAIL) * GA::PS; }
    static GIGA_HD int mid0(int R) { return WA + WB + imgA_bytes(R); }                    // first byte of B's image
    static GIGA_HD int lds_bytes(int R) { return mid0(R) + ((R + 2) * GB::P + C32_TAIL) * GB::PS; }
    static GIGA_HD void sub_bands(int sA, int sB, int& n, int& rows) {
        const int r = sB - sA;
        n = (r + RBMAX - 1) / RBMAX;
        rows = n ? (r + n - 1) / n : 0;
    }
    // LDS byte offset (from the start of B's image) of the 8-channel group `v` of A's output pixel (row orow of A's sub-band, column x)
    static GIGA_HD int mid_off(int orow, int x, int v) { return (orow * GB::P + x + 1) * GB::PS + v * GB::ILB; }
};

// The U-Net on conv32.  X(layer, KIND, C0, C1, COUT, H, W, POOLIN, SPW, SGM f16 / bf16, SGM f16x3, KP f16x3): layer order of
// giga_layout.h::kConv.  SPW = slices of a wave's register tile; SGM = slice groups among the members of a group, chosen so that
// a member's weights (SPM slices x taps x CIN / 16 KiB; twice that in the split mode) leave room for the image in 159 KiB; the
// two 128-input-channel layers of the split mode (144 KiB for ONE slice) walk their channels in two parts.
// Layers 2 and 4 read the un-pooled skip tensors and pool them while staging (and write the pooled tensors Q0 / Q1, which the
// workspace layout of the C ABI exposes).
#define GIGA_UNET32_LAYERS(X)                            \
    X(0, CONV3, 32, 0, 32, 40, 40, false, 1, 1, 1, 1)    \
    X(1, CONV3, 32, 0, 32, 40, 40, false, 1, 1, 1, 1)    \
    X(2, CONV3, 32, 0, 64, 20, 20, true, 2, 1, 1, 1)     \
    X(3, CONV3, 64, 0, 64, 20, 20, false, 2, 1, 2, 1)    \
    X(4, CONV3, 64, 0, 128, 10, 10, true, 2, 2, 4, 1)    \
    X(5, CONV3, 128, 0, 128, 10, 10, false, 1, 4, 4, 2)  \
    X(6, UPCONV, 128, 0, 64, 10, 10, false, 2, 1, 2, 1)  \
    X(7, CONV3, 64, 64, 64, 20, 20, false, 1, 2, 2, 2)   \
    X(8, CONV3, 64, 0, 64, 20, 20, false, 2, 1, 2, 1)    \
    X(9, UPCONV, 64, 0, 32, 20, 20, false, 2, 1, 1, 1)   \
    X(10, CONV3, 32, 32, 32, 40, 40, false, 1, 1, 1, 1)  \
    X(11, CONV3, 32, 0, 32, 40, 40, false, 1, 1, 1, 1)   \
    X(12, CONV1, 32, 0, 32, 40, 40, false, 1, 1, 1, 1)
template <int MODE, int L> struct U32Layer;
#define X(l, KIND, C0, C1, COUT, H, W, POOLIN, SPW, SGN, SGS, KPS)                                                \
    template <int MODE> struct U32Layer<MODE, l> {                                                                \
        using G = C32<MODE, KIND, C0, C1, COUT, H, W, POOLIN, SPW, (MODE == C32_SPLIT ? SGS : SGN),               \
                      (MODE == C32_SPLIT ? KPS : 1)>;                                                             \
        static constexpr bool RELU = KIND == CONV3;                                                               \
    };
GIGA_UNET32_LAYERS(X)
#undef X

}  // namespace giga
