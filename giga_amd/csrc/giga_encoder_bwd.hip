// Encoder backward (training path, fp32): gradients of every LocalVoxelEncoder / UNet parameter
// (reference: autograd through ConvONets/encoder/voxels.py:89-121 and encoder/unet.py:225-239, driven by
// scripts/train_giga.py:198-211).
//   * data gradients of the convolutions reuse conv16_kernel with flipped/transposed weight fragments
//     (backward blob, giga_pack.cpp::pack_conv_dgrad); ConvTranspose2d's data gradient is the DOWN kind
//   * weight gradients: conv_wgrad_kernel, an MFMA reduction over pixels D[co][ci] += dY[p][co] * X[p+tap][ci]
//   * ReLU / max-pool / concat backward: small elementwise kernels on the saved NHWC activations
//   * conv_in: recompute the pre-activation (never stored by the forward), mask, and reduce
//     dW[c][tap] = sum_voxels dF[c][v] * tsdf[v + tap] on the MFMA as well.
// Weight/bias gradients are ACCUMULATED into the flat fp32 gradient buffer (reference state-dict order) with
// fp32 atomics (summation order is not fixed, as with PyTorch's own CUDA/HIP backward kernels).
#include "giga_conv16.h"
#include "giga_wino.h"
#include <functional>
#include <mutex>

#include "giga_bwd_mega.h"
#include "giga_side.h"
#include "giga_args.h"

namespace giga {

// ------------------------------- elementwise ------------------------------------------------------------
// dS[img][y][x][c] = dCat[img][y][x][coff + c] (skip half of the concat gradient, channel stride cs)
//                  + (S[y][x][c] == Q[y/2][x/2][c] ? dQ[y/2][x/2][c] : 0)      (MaxPool2d(2,2) backward)
// and, S being the ReLU output of the layer below, its backward in the same pass: dS = S > 0 ? dS : 0
// (one thread per FOUR channels: 16-byte loads and stores; C, cs and coff are multiples of 4)
__global__ void pool_bwd_add_kernel(float* __restrict__ dS, const float* __restrict__ dcat, int cs, int coff,
                                    const float* __restrict__ dQ, const float* __restrict__ S,
                                    const float* __restrict__ Q, int nimg, int H, int W, int C) {
    const int C4 = C >> 2;
    const size_t total = (size_t)nimg * H * W * C4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = 4 * (int)(i % C4);
    const size_t pix = i / C4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const size_t img = pix / ((size_t)W * H);
    const size_t qi = ((img * (H / 2) + y / 2) * (W / 2) + x / 2) * C + c;
    const float4 s = *reinterpret_cast<const float4*>(S + pix * C + c);
    const float4 d = *reinterpret_cast<const float4*>(dcat + pix * cs + coff + c);
    const float4 q = *reinterpret_cast<const float4*>(Q + qi);
    const float4 g = *reinterpret_cast<const float4*>(dQ + qi);
    float4 o;
    o.x = s.x > 0.f ? d.x + (s.x == q.x ? g.x : 0.f) : 0.f;
    o.y = s.y > 0.f ? d.y + (s.y == q.y ? g.y : 0.f) : 0.f;
    o.z = s.z > 0.f ? d.z + (s.z == q.z ? g.z : 0.f) : 0.f;
    o.w = s.w > 0.f ? d.w + (s.w == q.w ? g.w : 0.f) : 0.f;
    *reinterpret_cast<float4*>(dS + pix * C + c) = o;
}

// ------------------------------- weight gradient (MFMA reduction over pixels) ----------------------------
// D[m][n] += sum_p R[p][m] * Cc[map(p, tap)][n]   on v_mfma_f32_32x32x2_f32 (2 pixels per MFMA).
//   CONV3 / CONV1 : R = dPre (m = co), Cc = layer input at the tap-shifted pixel (n = ci, zero outside)
//   UPCONV        : R = layer input (m = ci), Cc = dU at (2y+dy, 2x+dx) (n = co)
// Operands come straight from HBM/L2 (32 lanes x 4 B contiguous per pixel), no LDS staging: one MFMA
// (64 cycles) per two 128-byte loads.  Block = (tap, 32x32 block of dW, pixel range); its 16 waves split
// the range, partials are summed through LDS and added to the gradient buffer with one atomic per element.
// ~256 workgroups per launch: these launches are bound by their ATOMICS, not by the reduction (float atomics are fabric
// operations, and all pixel ranges of a block hit the same 1024 addresses): with 768 workgroups of 4 waves the 1x1 layer's
// gradient took 39 us and the two ConvTranspose layers' 31 us each for 4-5 us of loads and MFMAs; a third of the workgroups
// with four times the waves issue a third of the atomics and keep the same number of loads in flight.
struct WgradArgs {
    const float* R; int csR, coR;                 // row tensor, channel stride, first channel
    const float* C0p; const float* C1p;           // column tensor(s) (concat order), channel strides / split
    int csC0, csC1, nC0;                          // columns n < nC0 come from C0p, the rest from C1p
    float* dW; int sM, sN, sT;                    // dW[m*sM + n*sN + tap*sT]
    int kind, taps, Mb, Nb;                       // 32-row / 32-column blocks
    int nimg, H, W;                               // base pixel grid (the R tensor's)
    unsigned mHW, mW;                             // magic multipliers for / (H*W) and / W
    long long npix; int pix_per_block;
    float* partial;                               // [workgroup][16 registers][64 lanes]: the workgroups' 32x32 tiles, summed by conv_wgrad_reduce_kernel;
                                                  // behind them [workgroup][32]: column sums of the gradient tensor (bias gradient)
    float* db; int nbias;                         // bias gradient (cout entries) or nullptr
};

constexpr int WG_NW = 16;                          // waves per workgroup of conv_wgrad_kernel
template <bool UP>                                 // UP: the ConvTranspose form (a.kind == UPCONV); else CONV3 / CONV1
__global__ __launch_bounds__(WG_NW * 64) void conv_wgrad_kernel(WgradArgs a) {
    __shared__ float red[WG_NW][16][64];             // [wave][register][lane]: conflict-free both ways, 64 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, hi = lane >> 5;
    int b = blockIdx.x;
    const int nb = b % a.Nb; b /= a.Nb;
    const int mb = b % a.Mb; b /= a.Mb;
    const int tap = b % a.taps; b /= a.taps;
    const long long p0 = (long long)b * a.pix_per_block;
    long long p1 = p0 + a.pix_per_block;
    if (p1 > a.npix) p1 = a.npix;
    const int n0 = nb * 32;
    const float* Cc = n0 < a.nC0 ? a.C0p : a.C1p;
    const int csC = n0 < a.nC0 ? a.csC0 : a.csC1;
    const int cn = (n0 < a.nC0 ? n0 : n0 - a.nC0) + i;
    const int HW = a.H * a.W;
    const int ky = !UP && a.kind == CONV3 ? tap / 3 - 1 : 0, kx = !UP && a.kind == CONV3 ? tap % 3 - 1 : 0;

    // fp32 MFMA shares the VALU, so the loop keeps per-load vector work at zero: W is even, hence a pixel pair
    // (p even, p+1) never straddles a row and the pair's (image, row, column) are wave-uniform SALU values; the
    // per-lane part of every address (odd pixel of the pair for the upper lane half, channel) is loop-invariant.
    // The SCALAR work matters just as much: a CU has one scalar unit for its sixteen waves, and the first version of this loop
    // (two magic divisions and 64-bit index arithmetic per pixel pair, ~80 SALU instructions) made these launches 21-39 us for
    // ~4 us of loads and MFMAs.  Now: one division per step of four pairs, the pairs after the first by increment, 32-bit
    // element offsets from the tensor bases (launch_wgrad bounds the tensors), all eight loads of a step issued before the
    // first use.
    const unsigned laneR = (unsigned)(hi * a.csR + a.coR + mb * 32 + i);
    const unsigned laneC = (unsigned)((UP ? 2 * hi : hi) * csC + cn);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    // this wave's share: pixels p0 + 8*wave + 8*WG_NW*k + {0..7}; lane half hi takes the odd pixel of each pair
    for (long long base = p0 + 8 * wave; base < p1; base += 8 * WG_NW) {
        int img = div_magic((int)base, a.mHW), rem = (int)base - img * HW;
        int py = div_magic(rem, a.mW), px = rem - py * a.W;
        unsigned oR[4], oC[4];
        bool okR[4], okC[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            okR[s] = base + 2 * s < p1;                           // npix and the block ranges are even
            oR[s] = (unsigned)((int)base + 2 * s) * (unsigned)a.csR + laneR;
            if constexpr (UP) {
                const unsigned q = (unsigned)((img * 2 * a.H + 2 * py + (tap >> 1)) * (2 * a.W) + 2 * px + (tap & 1));
                oC[s] = q * (unsigned)csC + laneC; okC[s] = okR[s];
            } else {
                const int qy = py + ky, qx = px + kx;             // column of the even pixel's tap
                const bool row_in = qy >= 0 && qy < a.H;
                const bool lo_ok = qx >= 0, hi_ok = qx + 1 < a.W;           // uniform; qx+1 >= 0 and qx < W always
                okC[s] = okR[s] && row_in && (hi ? hi_ok : lo_ok);
                oC[s] = (unsigned)(img * HW + qy * a.W + qx) * (unsigned)csC + laneC;
            }
            if (!okR[s]) oR[s] = oR[0];                           // (clamped: the value is zeroed below)
            if (!okC[s]) oC[s] = laneC;
            px += 2;
            if (px >= a.W) { px = 0; if (++py >= a.H) { py = 0; ++img; } }
        }
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { av[s] = a.R[oR[s]]; bv[s] = Cc[oC[s]]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) { av[s] = okR[s] ? av[s] : 0.f; bv[s] = okC[s] ? bv[s] : 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) bsum += UP ? bv[s] : av[s];   // the gradient tensor's channel of this lane (bias gradient)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma32(av[s], bv[s], acc);
    }
    // D: lane (n = lane&31, hi), reg r <-> row m = drow(r, hi), column n.  Every wave parks its tile; wave w then sums register w
    // of all sixteen tiles (sixteen LDS reads per wave, side by side -- one wave summing fifteen tiles took longer than the loop).
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    static_assert(WG_NW == 16, "one wave per accumulator register");
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WG_NW; ++w) v += red[w][wave][lane];
    a.partial[((size_t)blockIdx.x * 16 + wave) * 64 + lane] = v;
    // bias gradient = column sums of the gradient tensor (dPre, or dU for the ConvTranspose form): the blocks that see every
    // channel exactly once per pixel range hand over their sums (UP: column block nb of every tap, first row block; else: row
    // block mb, first column block and tap)
    if (a.db && (UP ? mb == 0 : (nb == 0 && tap == 0))) {
        bsum += __shfl_xor(bsum, 32);
        __syncthreads();                                         // the tiles in `red` have been read
        if (hi == 0) red[wave][0][i] = bsum;
        __syncthreads();
        if (wave == 0 && hi == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WG_NW; ++w) t += red[w][0][i];
            a.partial[(size_t)gridDim.x * 1024 + (size_t)blockIdx.x * 32 + i] = t;
        }
    }
}

// dW[m][n][tap] += sum over the pixel ranges of the workgroups' tiles.  One workgroup per 64 tile elements: four quarter sums
// per element (every thread a quarter of the ranges, eight loads in flight), folded through LDS; every dW element has exactly
// one writer.  (Atomics instead -- 256...768 workgroups adding onto the same 1024 addresses per block -- made these launches
// 27-39 us for 4-5 us of loads and MFMAs: float atomics are fabric operations.)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(WgradArgs a, int blocks_wn, int ksplit) {
    __shared__ float part[4][64];
    if ((int)blockIdx.x >= blocks_wn * 16) {                     // the workgroups behind the tiles: bias gradient, 32 channels each
        const int cb = (int)blockIdx.x - blocks_wn * 16, i = threadIdx.x & 31, q8 = threadIdx.x >> 5;   // 8 partial sums per channel
        const bool up = a.kind == UPCONV;
        const float* bsrc = a.partial + (size_t)blocks_wn * ksplit * 1024;
        // contributing blocks: UP: (tap 0..3, mb = 0, nb = cb); else: (tap 0, mb = cb, nb = 0); index ((ks*taps + tap)*Mb + mb)*Nb + nb
        const int ntap = up ? a.taps : 1;
        float t = 0.f;
        for (int j = q8; j < ksplit * ntap; j += 8) {
            const int ks = j / ntap, tap = j - ks * ntap;
            const int blk = up ? ((ks * a.taps + tap) * a.Mb) * a.Nb + cb : ((ks * a.taps) * a.Mb + cb) * a.Nb;
            t += bsrc[(size_t)blk * 32 + i];
        }
        part[q8 >> 1][(q8 & 1) * 32 + i] = t;
        __syncthreads();
        if (threadIdx.x < 32) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += part[k >> 1][(k & 1) * 32 + i];
            a.db[cb * 32 + i] += v;
        }
        return;
    }
    const int blk = blockIdx.x >> 4, e = ((blockIdx.x & 15) << 6) + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const float* src = a.partial + (size_t)blk * 1024 + e;
    const size_t stride = (size_t)blocks_wn * 1024;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int ks = q;
    for (; ks + 12 < ksplit; ks += 16) {
        s0 += src[(size_t)ks * stride]; s1 += src[(size_t)(ks + 4) * stride];
        s2 += src[(size_t)(ks + 8) * stride]; s3 += src[(size_t)(ks + 12) * stride];
    }
    for (; ks < ksplit; ks += 4) s0 += src[(size_t)ks * stride];
    part[q][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0) {
        const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        int b = blk;
        const int nb = b % a.Nb; b /= a.Nb;
        const int mb = b % a.Mb; b /= a.Mb;
        const int tap = b;
        const int r = e >> 6, lane = e & 63, i = lane & 31, hi = lane >> 5;
        const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, n = nb * 32 + i;
        a.dW[(size_t)m * a.sM + (size_t)n * a.sN + (size_t)tap * a.sT] += v;
    }
}

static int launch_wgrad(WgradArgs a, hipStream_t s) {
    a.npix = (long long)a.nimg * a.H * a.W;
    a.mHW = (unsigned)((0x100000000ULL + (unsigned)(a.H * a.W) - 1) / (unsigned)(a.H * a.W));
    a.mW = (unsigned)((0x100000000ULL + (unsigned)a.W - 1) / (unsigned)a.W);
    // conv_wgrad_kernel addresses its tensors with 32-bit element offsets
    const long long cpix = a.npix * (a.kind == UPCONV ? 4 : 1);
    if (a.npix >= (1ll << 31) || a.npix * a.csR >= (1ll << 31) || cpix * (a.csC0 > a.csC1 ? a.csC0 : a.csC1) >= (1ll << 31)) return -7;
    const int blocks_wn = a.taps * a.Mb * a.Nb;
    int ksplit = 256 / blocks_wn;      // few workgroups per gradient element: every atomic is a fabric operation
    if (ksplit < 1) ksplit = 1;
    long long ppb = (a.npix + ksplit - 1) / ksplit;
    ppb = (ppb + 31) / 32 * 32;
    if (ppb < 32) ppb = 32;
    a.pix_per_block = (int)ppb;
    ksplit = (int)((a.npix + ppb - 1) / ppb);
    if (a.kind == UPCONV) GIGA_LAUNCH(conv_wgrad_kernel<true>, dim3(blocks_wn * ksplit), dim3(WG_NW * 64), 0, s, a);
    else GIGA_LAUNCH(conv_wgrad_kernel<false>, dim3(blocks_wn * ksplit), dim3(WG_NW * 64), 0, s, a);
    GIGA_LAUNCH(conv_wgrad_reduce_kernel, dim3(blocks_wn * 16 + (a.db ? a.nbias / 32 : 0)), dim3(256), 0, s, a, blocks_wn, ksplit);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

// ------------------------------- 3x3 weight gradient, LDS-staged -------------------------------------------------
// dW[co][ci][tap] += sum_p dY[p][co] * X[p + tap][ci] for one 3x3 layer.  Unit = (image, strip of RS rows): the
// workgroup stages the haloed input strip (all input channels, zero outside the image) and the dY strip (all
// output channels) in LDS ONCE; its 8 waves then take their operands from LDS (one ds_read_b32 per 64-cycle MFMA,
// immediate tap offsets, two VALU per pixel pair -- fp32 MFMA shares the VALU).  A wave owns one 32x32 (co, ci)
// block for ALL nine taps (nine 32x32 accumulators, the dY operand is read once per nine MFMAs) and one of KS
// interleaved shares of the strip's pixel pairs; blocks beyond 8 per workgroup go to blockIdx.y.  Accumulators
// live across the whole persistent strip loop; the next strip is prefetched into registers under the MFMAs.
// Epilogue: the waves of a block add their tiles into an LDS image laid out like the parameter ([co][ci][tap],
// ds_add_f32), which is then added to the gradient with fully coalesced atomics.
struct Wgrad3Args {
    const float* dY;                 // [img][H][W][COUT]   gradient of the layer's pre-activation
    const float* in0; const float* in1;   // layer input (concat order), dense NHWC with C0 / C1 channels
    float* dW;                       // [COUT][CIN][3][3], accumulated by wgrad3_reduce_kernel
    float* db;                       // [COUT] bias gradient = column sums of dY, accumulated by wgrad3_reduce_kernel
    float* partial;                  // per-workgroup partial sums (workspace): [by][bx][block][9][32][32], then [bx][COUT]
    int nimg;
};
constexpr int WG3_MAX_PARTS = 256 * 8;       // (workgroup, block) partial images of 9x32x32 floats
constexpr int WG3_BIAS_FLOATS = 256 * 128;   // per-workgroup bias partials behind them

#ifdef GIGA_TRACE
static __device__ long long g_wg3_trace[8 * 32];
#define WG3_T(idx) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (idx) < 32) \
        g_wg3_trace[wave * 32 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG3_T(idx) do {} while (0)
#endif

template <int C0, int C1, int COUT, int H, int RS>
__global__ __launch_bounds__(512) void conv3_wgrad_kernel(Wgrad3Args a) {
    constexpr int W = H, CIN = C0 + C1;
    constexpr int NBK = CIN / 32, NBLK = (COUT / 32) * NBK;
    constexpr int BPG = NBLK < 8 ? NBLK : 8;          // blocks per workgroup
    constexpr int KS = 8 / BPG;                       // waves sharing a block (K split)
    constexpr int XS = CIN + (CIN % 64 == 0 ? 32 : 0);      // LDS pixel strides (floats): odd multiple of 32 so that
    constexpr int YS = COUT + (COUT % 64 == 0 ? 32 : 0);    // the two pixels of a pair fall into different bank halves
    constexpr int XW = W + 2, XR = RS + 2;
    constexpr int NVX = XR * XW * (CIN / 4), NVY = RS * W * (COUT / 4), NV = NVX + NVY;
    constexpr int NLD = (NV + 511) / 512;
    constexpr int XBYTES = XR * XW * XS * 4;
    constexpr int NPAIR = RS * W / 2;
    static_assert(H % RS == 0 && W % 2 == 0, "strip geometry");
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    float* Xs = wg_lds;
    float* Ys = wg_lds + XR * XW * XS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, k = lane >> 5;
    const int blk = blockIdx.y * BPG + wave / KS, ks = wave % KS;
    const int mb = blk / NBK, nb = blk % NBK;

    // staging geometry of this thread's vectors (fixed for the whole kernel): LDS float offset and source coordinates
    int st_lds[NLD], st_src[NLD];                     // src: X: row<<20 | col<<10 | channel ; dY: 1<<30 | pixel<<10 | channel
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int v = tid + 512 * q;
        if (v < NVX) {
            const int pix = v / (CIN / 4), c = (v % (CIN / 4)) * 4;
            st_lds[q] = pix * XS + c;
            st_src[q] = ((pix / XW) << 20) | ((pix % XW) << 10) | c;
        } else if (v < NV) {
            const int u = v - NVX, pix = u / (COUT / 4), c = (u % (COUT / 4)) * 4;
            st_lds[q] = XR * XW * XS + pix * YS + c;
            st_src[q] = (1 << 30) | (pix << 10) | c;
        } else {
            st_lds[q] = -1; st_src[q] = 0;
        }
    }
    constexpr int SPI = H / RS;                       // strips per image
    const int nstrips = a.nimg * SPI;
    float4 stg[NLD];
    auto issue = [&](int st) {
        const int img = st / SPI, y0 = (st % SPI) * RS;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            const int sd = st_src[q], c = sd & 1023;
            if (st_lds[q] >= 0) {
                if (sd >> 30) {
                    const int pix = (sd >> 10) & 0xFFFFF;
                    val = *reinterpret_cast<const float4*>(a.dY + ((size_t)(img * H + y0) * W + pix) * COUT + c);
                } else {
                    const int gy = y0 - 1 + (sd >> 20), gx = ((sd >> 10) & 1023) - 1;
                    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                        const size_t px = (size_t)(img * H + gy) * W + gx;
                        val = c < C0 ? *reinterpret_cast<const float4*>(a.in0 + px * C0 + c)
                                     : *reinterpret_cast<const float4*>(a.in1 + px * C1 + (c - C0));
                    }
                }
            }
            stg[q] = val;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane_a = k * YS + mb * 32 + i;          // dY operand: row = co, k = pixel of the pair
    const int lane_b = k * XS + nb * 32 + i;          // X operand: column = ci

    WG3_T(0);
    int tc = 1;
    float bsum = 0.f;
    int st = blockIdx.x;
    if (st < nstrips) issue(st);
    for (; st < nstrips; st += gridDim.x) {
        __syncthreads();                              // everyone is done reading the previous strip
#pragma unroll
        for (int q = 0; q < NLD; ++q)
            if (st_lds[q] >= 0) *reinterpret_cast<float4*>(wg_lds + st_lds[q]) = stg[q];
        __syncthreads();
        WG3_T(tc); ++tc;
        if (st + (int)gridDim.x < nstrips) issue(st + gridDim.x);      // flies under this strip's MFMAs
        WG3_T(tc); ++tc;
        // bias gradient: column sums of the staged dY strip (ten adds per thread per strip)
        if (blockIdx.y == 0) {
            const int bc = tid % COUT;
            for (int px = tid / COUT; px < RS * W; px += 512 / COUT) bsum += Ys[px * YS + bc];
        }
        // this wave's pixel pairs ks, ks+KS, ...: the ten operands of pair j+1 are read from LDS before the nine
        // MFMAs of pair j are issued (ping-pong registers), so the LDS round trip never stalls the MFMA pipe
        constexpr int CNT = NPAIR / KS;
        static_assert(NPAIR % KS == 0 && CNT % 2 == 0, "pair split");
        auto ld = [&](int j, float& av, float (&bv)[9]) {
            const int pi = ks + j * KS;
            const int r = pi / (W / 2), x = 2 * (pi % (W / 2));
            av = Ys[(r * W + x) * YS + lane_a];
            const float* xb = Xs + (r * XW + x) * XS + lane_b;
#pragma unroll
            for (int t = 0; t < 9; ++t) bv[t] = xb[((t / 3) * XW + (t % 3)) * XS];
        };
        float a0, b0[9], a1, b1[9];
        ld(0, a0, b0);
        for (int j = 0; j < CNT; j += 2) {
            ld(j + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = mfma32(a0, b0[t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < CNT) ld(j + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = mfma32(a1, b1[t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        WG3_T(tc); ++tc;
    }
    WG3_T(30);
    // ---- epilogue: three rounds of three taps.  Every wave parks its three 32x32 tiles in its own LDS slot (register
    // layout, 16-B vectors, conflict-free), then all 512 threads sum the KS slots of every block and store the result
    // tap-major ([block][tap][co][ci], coalesced) into this workgroup's partial image; wgrad3_reduce_kernel sums the
    // workgroups and transposes to the parameter layout.  (Device-scope float atomics from 256 workgroups onto one
    // 36 KiB image, and LDS float atomics, each cost several times the MFMA loop.)
    if (blockIdx.y == 0) {                                            // fold the 512/COUT pixel shares of every channel
        __syncthreads();
        wg_lds[tid] = bsum;
        __syncthreads();
        if (tid < COUT) {
            float sb = 0.f;
            for (int q = 0; q < 512 / COUT; ++q) sb += wg_lds[q * COUT + tid];
            a.partial[(size_t)gridDim.y * gridDim.x * BPG * 9 * 1024 + (size_t)blockIdx.x * COUT + tid] = sb;
        }
    }
    float4* slot = reinterpret_cast<float4*>(wg_lds);                 // [wave][tap of the round][r4][lane]
    constexpr int RN = 9 * 1024;
    float* part = a.partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BPG) * RN;
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < 3; ++tl)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                slot[((wave * 3 + tl) * 4 + r4) * 64 + lane] =
                    make_float4(acc[3 * g3 + tl][4 * r4], acc[3 * g3 + tl][4 * r4 + 1], acc[3 * g3 + tl][4 * r4 + 2],
                                acc[3 * g3 + tl][4 * r4 + 3]);
        __syncthreads();
        for (int v = tid; v < BPG * 3 * 4 * 64; v += 512) {
            const int ln = v & 63, r4 = (v >> 6) & 3, tl = (v >> 8) % 3, b = v / 768;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const float4 x = slot[(((b * KS + q) * 3 + tl) * 4 + r4) * 64 + ln];
                sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
            }
            const int n = ln & 31, m0 = 8 * r4 + 4 * (ln >> 5);       // D rows of registers 4*r4 .. 4*r4+3: m0 .. m0+3
            float* dst = part + ((size_t)(b * 9 + 3 * g3 + tl) * 32 + m0) * 32 + n;
            dst[0] = sum.x; dst[32] = sum.y; dst[64] = sum.z; dst[96] = sum.w;
        }
    }
    WG3_T(31);
}

// dW[co][ci][tap] of block gb = sum over the nx workgroups of partial[by][bx][b][tap][co][ci]   (gb = by*BPG + b).
// A workgroup takes 64 elements of a block; every element's sum over the nx partial images is cut into four interleaved quarter
// sums (one per wave, up to eight loads in flight), folded through LDS; every dW element has exactly one writer.  (Round 2: one
// thread per element walked all nx = 128-256 images -- a chain of 32-64 dependent round trips, 13-24 us per layer, 128 us of the
// bf16 step -- or, for the small layers, NZ groups meeting in dW with atomics.)
template <int CIN, int BPG>
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const float* __restrict__ partial, int nx, float* __restrict__ dW,
                                                            float* __restrict__ db, int cout, int ny) {
    constexpr int RN = 9 * 1024, NBK = CIN / 32;
    __shared__ float sb[256];
    if (blockIdx.x == RN / 64) {                                // one extra block: the bias gradient
        if (blockIdx.y != 0) return;
        const int c = threadIdx.x % cout, part = threadIdx.x / cout, np = 256 / cout;     // cout divides 256
        const float* bsrc = partial + (size_t)ny * nx * BPG * RN + c;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int x = part;
        for (; x + 3 * np < nx; x += 4 * np) {
            v0 += bsrc[(size_t)x * cout]; v1 += bsrc[(size_t)(x + np) * cout];
            v2 += bsrc[(size_t)(x + 2 * np) * cout]; v3 += bsrc[(size_t)(x + 3 * np) * cout];
        }
        for (; x < nx; x += np) v0 += bsrc[(size_t)x * cout];
        sb[threadIdx.x] = (v0 + v1) + (v2 + v3);
        __syncthreads();
        if ((int)threadIdx.x < cout) {
            float t = 0.f;
            for (int q = 0; q < np; ++q) t += sb[q * cout + threadIdx.x];
            db[threadIdx.x] += t;
        }
        return;
    }
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int gb = blockIdx.y, by = gb / BPG, b = gb % BPG;
    const float* src = partial + ((size_t)by * nx * BPG + b) * RN + e;
    constexpr size_t ST = (size_t)BPG * RN;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    int x = q;
    for (; x + 28 < nx; x += 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += src[(size_t)(x + 4 * k) * ST];
    }
    for (; x < nx; x += 4) s[0] += src[(size_t)x * ST];
    sb[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (q == 0) {
        const float v = (sb[threadIdx.x] + sb[64 + threadIdx.x]) + (sb[128 + threadIdx.x] + sb[192 + threadIdx.x]);
        const int t = e >> 10, m = (e >> 5) & 31, n = e & 31;
        const int gmb = gb / NBK, gnb = gb % NBK;
        dW[((size_t)(gmb * 32 + m) * CIN + gnb * 32 + n) * 9 + t] += v;
    }
}

// ONE reduce launch for the 3x3 layers of a backward pass (round 5): every layer's weight-gradient launch leaves its partial images in
// its OWN region of the workspace and a descriptor here; this kernel runs the blocks of wgrad3_reduce_kernel for all of them (same
// element -> thread map, same order of summation: bit-identical results) in one grid.  Ten launches of 145 x NBLK blocks -- 9-13 us
// each, none of them filling the chip -- become one of ~6 800 blocks.
struct Wg3RedLayer {
    const float* partial; float* dW; float* db;
    int nx, ny, bpg, cin, cout, first_block;      // first_block: prefix sum of (144 * nblk + 1) over the layers before this one
};
constexpr int WG3_RED_MAX = 10;
struct Wg3RedArgs { Wg3RedLayer L[WG3_RED_MAX]; int nlayers; int nblocks; };

__global__ __launch_bounds__(256) void wgrad3_reduce_all_kernel(Wg3RedArgs A) {
    constexpr int RN = 9 * 1024;
    __shared__ float sb[256];
    const int bid = blockIdx.x;
    const float* partial = A.L[0].partial; float* dW = A.L[0].dW; float* db = A.L[0].db;
    int nx = A.L[0].nx, ny = A.L[0].ny, bpg = A.L[0].bpg, cin = A.L[0].cin, cout = A.L[0].cout, first = 0;
#pragma unroll
    for (int i = 1; i < WG3_RED_MAX; ++i)
        if (i < A.nlayers && bid >= A.L[i].first_block) {
            partial = A.L[i].partial; dW = A.L[i].dW; db = A.L[i].db;
            nx = A.L[i].nx; ny = A.L[i].ny; bpg = A.L[i].bpg; cin = A.L[i].cin; cout = A.L[i].cout; first = A.L[i].first_block;
        }
    const int local = bid - first, nbk = cin >> 5, nblk = (cout >> 5) * nbk;
    if (local == 144 * nblk) {                                  // the layer's bias gradient
        const int c = threadIdx.x % cout, part = threadIdx.x / cout, np = 256 / cout;     // cout divides 256
        const float* bsrc = partial + (size_t)ny * nx * bpg * RN + c;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int x = part;
        for (; x + 3 * np < nx; x += 4 * np) {
            v0 += bsrc[(size_t)x * cout]; v1 += bsrc[(size_t)(x + np) * cout];
            v2 += bsrc[(size_t)(x + 2 * np) * cout]; v3 += bsrc[(size_t)(x + 3 * np) * cout];
        }
        for (; x < nx; x += np) v0 += bsrc[(size_t)x * cout];
        sb[threadIdx.x] = (v0 + v1) + (v2 + v3);
        __syncthreads();
        if ((int)threadIdx.x < cout) {
            float t = 0.f;
            for (int q = 0; q < np; ++q) t += sb[q * cout + threadIdx.x];
            db[threadIdx.x] += t;
        }
        return;
    }
    const int gb = local / 144, ex = local - gb * 144;
    const int e = ex * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int by = gb / bpg, b = gb - by * bpg;
    const float* src = partial + ((size_t)by * nx * bpg + b) * RN + e;
    const size_t ST = (size_t)bpg * RN;
    float sacc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sacc[k] = 0.f;
    int x = q;
    for (; x + 28 < nx; x += 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc[k] += src[(size_t)(x + 4 * k) * ST];
    }
    for (; x < nx; x += 4) sacc[0] += src[(size_t)x * ST];
    sb[threadIdx.x] = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + ((sacc[4] + sacc[5]) + (sacc[6] + sacc[7]));
    __syncthreads();
    if (q == 0) {
        const float v = (sb[threadIdx.x] + sb[64 + threadIdx.x]) + (sb[128 + threadIdx.x] + sb[192 + threadIdx.x]);
        const int t = e >> 10, m = (e >> 5) & 31, n = e & 31;
        const int gmb = gb / nbk, gnb = gb - gmb * nbk;
        dW[((size_t)(gmb * 32 + m) * cin + gnb * 32 + n) * 9 + t] += v;
    }
}

// what a layer's launcher does with its reduce: launch it (defer == nullptr) or describe it for wgrad3_reduce_all_kernel
static void wg3_defer(Wg3RedArgs* R, const float* partial, int nx, int ny, int bpg, int cin, int cout, float* dW, float* db) {
    Wg3RedLayer& L = R->L[R->nlayers++];
    L.partial = partial; L.dW = dW; L.db = db; L.nx = nx; L.ny = ny; L.bpg = bpg; L.cin = cin; L.cout = cout;
    L.first_block = R->nblocks;
    R->nblocks += 144 * (cout / 32) * (cin / 32) + 1;
}
static int launch_wgrad3_reduce_all(const Wg3RedArgs& R, hipStream_t s) {
    if (R.nlayers == 0) return 0;
    GIGA_LAUNCH(wgrad3_reduce_all_kernel, dim3(R.nblocks), dim3(256), 0, s, R);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

template <int C0, int C1, int COUT, int H, int RS>
static int launch_wgrad3(const Wgrad3Args& a, hipStream_t s, Wg3RedArgs* defer = nullptr) {
    constexpr int CIN = C0 + C1, W = H;
    constexpr int NBLK = (COUT / 32) * (CIN / 32), BPG = NBLK < 8 ? NBLK : 8, NY = NBLK / BPG;
    constexpr int XS = CIN + (CIN % 64 == 0 ? 32 : 0), YS = COUT + (COUT % 64 == 0 ? 32 : 0);
    constexpr size_t strip = ((size_t)(RS + 2) * (W + 2) * XS + (size_t)RS * W * YS) * 4;
    constexpr size_t lds = strip > 8 * 3 * 4 * 64 * 16 ? strip : 8 * 3 * 4 * 64 * 16;      // strip or the epilogue slots (96 KiB)
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int nstrips = a.nimg * (H / RS);
    int gx = 256 / NY;
    if (gx > nstrips) gx = nstrips;
    auto kern = conv3_wgrad_kernel<C0, C1, COUT, H, RS>;
    giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
    GIGA_LAUNCH(kern, dim3(gx, NY), dim3(512), lds, s, a);
    if (defer) { wg3_defer(defer, a.partial, gx, NY, BPG, CIN, COUT, a.dW, a.db); return hipGetLastError() == hipSuccess ? 0 : -10; }
    GIGA_LAUNCH((wgrad3_reduce_kernel<CIN, BPG>), dim3(9 * 1024 / 64 + 1, NBLK), dim3(256), 0, s, a.partial, gx,
                       a.dW, a.db, COUT, NY);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

// ------------------------------- 3x3 weight gradient on bf16 MFMA (BASELINE c5) -----------------------------
// dW[co][ci][tap] = sum_pixels dY[p][co] * X[p + tap][ci]   with bf16 operands, fp32 accumulation: v_mfma_f32_32x32x16_bf16,
// D[co][ci] += A[co][k] * B[k][ci], k = 16 PIXELS per instruction (the fp32 kernel above: 2 pixels per 64-cycle instruction).
// Same decomposition, partial images, epilogue and reduce kernel as conv3_wgrad_kernel; what changes is the operand side:
//   * the contraction runs over pixels, so a lane needs 8 consecutive pixels of ONE channel: the strips are staged
//     TRANSPOSED, channel-major bf16 -- Xt[ci][row][col], Yt[co][row][col].  A staging item is (4 channels, row, 8-pixel
//     group): eight coalesced float4 loads (a pixel's channels are contiguous across lanes), converted and written as four
//     16-byte rows.  Rounding to bf16 happens here, once per element.
//   * a k-chunk = two 8-pixel groups (lane half hi takes group 2j + hi); groups tile the rows (40 = 5 groups; 20 and 10 are
//     padded to 24 / 16 with zeros, which contribute nothing).
//   * the three horizontal taps are ONE aligned 16-byte read plus the two neighbouring elements (two 4-byte reads), shifted
//     into place with v_alignbyte_b32: per k-chunk 1 + 3 x 3 LDS reads and 24 VALU feed nine MFMAs (288 matrix cycles).
//   X columns: data at col 8 + x; cols < 8 and >= 8 + 8G stay zero (the convolution's zero padding), zeroed once.
typedef __bf16 wbf8 __attribute__((ext_vector_type(8)));
// channel stride (bf16 elements) of the transposed strips: the smallest value >= n that is 8 mod 128, i.e. 16 bytes mod 256: the 32
// lanes of an operand read (one channel each, 16 bytes) then fall into 16 different 16-byte bank groups per pass -- with the natural
// stride (rows x pitch: 0 or 64 bytes mod 256 for every layer) they shared one to four (a third of the launch's cycles carried LDS
// bank conflicts: 2-8.6 M per launch) -- and the staging stores of 16 channel quads hit four bank groups instead of one.
constexpr int wg_cs(int n) { return (n - 8 + 127) / 128 * 128 + 8; }

template <int C0, int C1, int COUT, int H, int RS>
__global__ __launch_bounds__(512) void conv3_wgrad_bf16_kernel(Wgrad3Args a) {
    constexpr int W = H, CIN = C0 + C1;
    constexpr int NBK = CIN / 32, NBLK = (COUT / 32) * NBK;
    constexpr int BPG = NBLK < 8 ? NBLK : 8;          // blocks per workgroup
    constexpr int KS = 8 / BPG;                       // waves sharing a block (K split)
    constexpr int G = (W + 7) / 8;                    // 8-pixel groups per row
    constexpr int PY = 8 * G, PX = 8 * G + 16;        // row pitches (bf16 elements)
    constexpr int XR = RS + 2;
    constexpr int NVX = (CIN / 4) * XR * G, NVY = (COUT / 4) * RS * G, NV = NVX + NVY;
    static_assert(NV <= 512, "one staging item per thread");
    constexpr int XCS = wg_cs(XR * PX), YCS = wg_cs(RS * PY);        // channel strides
    constexpr int XELEMS = CIN * XCS, YELEMS = COUT * YCS;
    constexpr int NCH = RS * G / 2;                   // k-chunks (two groups each) per strip
    static_assert((RS * G) % 2 == 0 && H % RS == 0, "strip geometry");
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    __bf16* Xt = reinterpret_cast<__bf16*>(wg_lds);
    __bf16* Yt = Xt + XELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int blk = blockIdx.y * BPG + wave / KS, ks = wave % KS;
    const int mb = blk / NBK, nb = blk % NBK;

    // zero the whole X image once: the halo columns are never written again
    for (int v = tid; v < XELEMS / 8; v += 512) reinterpret_cast<uint4*>(Xt)[v] = make_uint4(0, 0, 0, 0);

    // this thread's staging item
    const bool isx = tid < NVX, isy = !isx && tid < NV;
    const int u = isx ? tid : tid - NVX;
    const int cq = isx ? u % (CIN / 4) : u % (COUT / 4);
    const int rest = isx ? u / (CIN / 4) : u / (COUT / 4);
    const int sg = rest % G, sr = rest / G;           // group within the row, strip row
    constexpr int SPI = H / RS;
    const int nstrips = a.nimg * SPI;
    // All eight loads of a staging item are issued UNCONDITIONALLY from clamped addresses and zeroed when they are stored
    // (`ok`): behind `if`s every load sat in its own basic block, the register allocator overlapped their destinations and the
    // compiler put s_waitcnt vmcnt(0) between them -- an item cost two to three dependent memory round trips instead of one
    // (the strip loop ran at ~6 k clocks per strip, tools/gpu_wgrad_trace.py).
    float4 stg[8];
    unsigned ok = 0;                                  // bit e: element e of the staged item is inside the image
    auto issue = [&](int st) {
        const int img = st / SPI, y0 = (st % SPI) * RS;
        const int gy = y0 - 1 + sr, gyc = gy < 0 ? 0 : gy >= H ? H - 1 : gy, c = 4 * cq;
        // (threads without an item read like a dY item of row 0: always a valid address)
        const float* rowp = !isx ? a.dY + ((size_t)(img * H + y0 + (isy ? sr : 0)) * W) * COUT + c
                                 : c < C0 ? a.in0 + ((size_t)(img * H + gyc) * W) * C0 + c
                                          : a.in1 + ((size_t)(img * H + gyc) * W) * C1 + (c - C0);
        const int ps = !isx ? COUT : c < C0 ? C0 : C1;
        const bool row_ok = isy || (isx && gy >= 0 && gy < H);
        ok = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int x = 8 * sg + e, xc = x < W ? x : W - 1;
            stg[e] = *reinterpret_cast<const float4*>(rowp + (size_t)xc * ps);
            ok |= (row_ok && x < W) ? 1u << e : 0u;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const __bf16* ya = Yt + (size_t)(mb * 32 + n) * YCS;              // this lane's dY channel row block
    const __bf16* xb = Xt + (size_t)(nb * 32 + n) * XCS + 8;          // this lane's X channel, col of x = 0

    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int st = blockIdx.x;
    if (st < nstrips) issue(st);
    for (; st < nstrips; st += gridDim.x) {
        __syncthreads();                              // everyone is done reading the previous strip (and the zero fill)
        const unsigned okc = ok;
        if (isx || isy) {
            __bf16* dst = isx ? Xt + (size_t)(4 * cq) * XCS + sr * PX + 8 + 8 * sg : Yt + (size_t)(4 * cq) * YCS + sr * PY + 8 * sg;
            const int cstride = isx ? XCS : YCS;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float v8[8] = {ch == 0 ? stg[0].x : ch == 1 ? stg[0].y : ch == 2 ? stg[0].z : stg[0].w,
                                     ch == 0 ? stg[1].x : ch == 1 ? stg[1].y : ch == 2 ? stg[1].z : stg[1].w,
                                     ch == 0 ? stg[2].x : ch == 1 ? stg[2].y : ch == 2 ? stg[2].z : stg[2].w,
                                     ch == 0 ? stg[3].x : ch == 1 ? stg[3].y : ch == 2 ? stg[3].z : stg[3].w,
                                     ch == 0 ? stg[4].x : ch == 1 ? stg[4].y : ch == 2 ? stg[4].z : stg[4].w,
                                     ch == 0 ? stg[5].x : ch == 1 ? stg[5].y : ch == 2 ? stg[5].z : stg[5].w,
                                     ch == 0 ? stg[6].x : ch == 1 ? stg[6].y : ch == 2 ? stg[6].z : stg[6].w,
                                     ch == 0 ? stg[7].x : ch == 1 ? stg[7].y : ch == 2 ? stg[7].z : stg[7].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (okc >> e & 1u) ? v8[e] : 0.f;
                wbf8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v8[e];
                *reinterpret_cast<wbf8*>(dst + (size_t)ch * cstride) = o;
                if (isy && blockIdx.y == 0)           // bias gradient: column sums of dY, from the fp32 values
                    bsum[ch] += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
            }
        }
        __syncthreads();
        if (st + (int)gridDim.x < nstrips) issue(st + gridDim.x);      // flies under this strip's MFMAs
        for (int j = ks; j < NCH; j += KS) {
            const int gi = 2 * j + hi, r = gi / G, xg = gi % G;
            const wbf8 A = *reinterpret_cast<const wbf8*>(ya + r * PY + 8 * xg);
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const __bf16* row = xb + (r + ty) * PX + 8 * xg;                 // X row y + ty - 1, col of the group's x0
                const uint4 mid = *reinterpret_cast<const uint4*>(row);
                const unsigned wp = *reinterpret_cast<const unsigned*>(row - 2);  // elements x0-2, x0-1
                const unsigned wn = *reinterpret_cast<const unsigned*>(row + 8);  // elements x0+8, x0+9
                const uint4 left = {__builtin_amdgcn_alignbyte(mid.x, wp, 2), __builtin_amdgcn_alignbyte(mid.y, mid.x, 2),
                                    __builtin_amdgcn_alignbyte(mid.z, mid.y, 2), __builtin_amdgcn_alignbyte(mid.w, mid.z, 2)};
                const uint4 right = {__builtin_amdgcn_alignbyte(mid.y, mid.x, 2), __builtin_amdgcn_alignbyte(mid.z, mid.y, 2),
                                     __builtin_amdgcn_alignbyte(mid.w, mid.z, 2), __builtin_amdgcn_alignbyte(wn, mid.w, 2)};
                acc[3 * ty + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(wbf8, left), acc[3 * ty + 0], 0, 0, 0);
                acc[3 * ty + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(wbf8, mid), acc[3 * ty + 1], 0, 0, 0);
                acc[3 * ty + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(wbf8, right), acc[3 * ty + 2], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: as conv3_wgrad_kernel (the D layout of the bf16 MFMA is the same 32x32 map) ----------------------
    if (blockIdx.y == 0) {
        __syncthreads();
        // per-channel totals: threads of the dY items hold 4 channels each for their (row, group)
        float* red = wg_lds;                                          // [COUT] after zeroing
        for (int c = tid; c < COUT; c += 512) red[c] = 0.f;
        __syncthreads();
        if (isy) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) atomicAdd(red + 4 * cq + ch, bsum[ch]);      // LDS, <= RS*G addends per channel
        }
        __syncthreads();
        if (tid < COUT)
            a.partial[(size_t)gridDim.y * gridDim.x * BPG * 9 * 1024 + (size_t)blockIdx.x * COUT + tid] = red[tid];
    }
    float4* slot = reinterpret_cast<float4*>(wg_lds);                 // [wave][tap of the round][r4][lane]
    constexpr int RN = 9 * 1024;
    float* part = a.partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BPG) * RN;
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < 3; ++tl)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                slot[((wave * 3 + tl) * 4 + r4) * 64 + lane] =
                    make_float4(acc[3 * g3 + tl][4 * r4], acc[3 * g3 + tl][4 * r4 + 1], acc[3 * g3 + tl][4 * r4 + 2],
                                acc[3 * g3 + tl][4 * r4 + 3]);
        __syncthreads();
        for (int v = tid; v < BPG * 3 * 4 * 64; v += 512) {
            const int ln = v & 63, r4 = (v >> 6) & 3, tl = (v >> 8) % 3, b = v / 768;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const float4 x = slot[(((b * KS + q) * 3 + tl) * 4 + r4) * 64 + ln];
                sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
            }
            const int nn = ln & 31, m0 = 8 * r4 + 4 * (ln >> 5);      // D rows of registers 4*r4 .. 4*r4+3: m0 .. m0+3
            float* dst = part + ((size_t)(b * 9 + 3 * g3 + tl) * 32 + m0) * 32 + nn;
            dst[0] = sum.x; dst[32] = sum.y; dst[64] = sum.z; dst[96] = sum.w;
        }
    }
}

// ---- the same kernel with the OUTPUT split over the workgroups as well (the layers with eight or sixteen 32x32 blocks) -------------
// In conv3_wgrad_bf16_kernel a workgroup computes eight blocks for its share of the strips and leaves a partial image of 8 x 36 KiB:
// with 256 workgroups that is 75 MB written and 75 MB read back by the reduce kernel PER LAYER (up0.conv1, down2.conv1, down2.conv2:
// 450 MB of the step's traffic for 7 MB of gradients) against 20-40 MB of operands.  Here a workgroup owns BX blocks that share
// their dY channels (one 32-channel block mb, the X blocks nb0 .. nb0 + BX - 1), the grid is (strips' share, all block groups), and
// only the channels those blocks need are staged: 32 of dY and 32 BX of X.  Same strips, k-chunks, MFMAs and epilogue; 8 / BX
// waves share a block (K split inside the workgroup).  Partial images: 256 workgroups x BX x 36 KiB = 19 MB at BX = 2.
template <int C0, int C1, int COUT, int H, int RS, int BX>
__global__ __launch_bounds__(512) void conv3_wgrad_bf16_ns_kernel(Wgrad3Args a) {
    constexpr int W = H, CIN = C0 + C1;
    constexpr int NBK = CIN / 32, NBLK = (COUT / 32) * NBK;
    constexpr int BPG = BX;                           // blocks per workgroup: (mb, nb0 .. nb0 + BX - 1)
    constexpr int KS = 8 / BPG;                       // waves sharing a block (K split)
    constexpr int NGX = NBK / BX;                     // X block groups per dY block
    static_assert(NBK % BX == 0 && 8 % BX == 0 && (C1 == 0 || C0 % (32 * BX) == 0), "block groups");
    constexpr int CINW = 32 * BX, COUTW = 32;         // channels this workgroup stages
    constexpr int G = (W + 7) / 8;                    // 8-pixel groups per row
    constexpr int PY = 8 * G, PX = 8 * G + 16;        // row pitches (bf16 elements)
    constexpr int XR = RS + 2;
    constexpr int NVX = (CINW / 4) * XR * G, NVY = (COUTW / 4) * RS * G, NV = NVX + NVY;
    static_assert(NV <= 512, "one staging item per thread");
    constexpr int XCS = wg_cs(XR * PX), YCS = wg_cs(RS * PY);        // channel strides
    constexpr int XELEMS = CINW * XCS, YELEMS = COUTW * YCS;
    constexpr int NCH = RS * G / 2;                   // k-chunks (two groups each) per strip
    static_assert((RS * G) % 2 == 0 && H % RS == 0, "strip geometry");
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    __bf16* Xt = reinterpret_cast<__bf16*>(wg_lds);
    __bf16* Yt = Xt + XELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int mb = blockIdx.y / NGX, nb0 = (blockIdx.y % NGX) * BX;       // this workgroup's dY block and first X block
    const int nbl = wave / KS, ks = wave % KS;                            // this wave's block (local X block index) and K share

    // zero the whole X image once: the halo columns are never written again
    for (int v = tid; v < XELEMS / 8; v += 512) reinterpret_cast<uint4*>(Xt)[v] = make_uint4(0, 0, 0, 0);

    // this thread's staging item
    const bool isx = tid < NVX, isy = !isx && tid < NV;
    const int u = isx ? tid : tid - NVX;
    const int cq = isx ? u % (CINW / 4) : u % (COUTW / 4);       // channel quad inside the staged channels
    const int rest = isx ? u / (CINW / 4) : u / (COUTW / 4);
    const int sg = rest % G, sr = rest / G;           // group within the row, strip row
    constexpr int SPI = H / RS;
    const int nstrips = a.nimg * SPI;
    // All eight loads of a staging item are issued UNCONDITIONALLY from clamped addresses and zeroed when they are stored
    // (`ok`): behind `if`s every load sat in its own basic block, the register allocator overlapped their destinations and the
    // compiler put s_waitcnt vmcnt(0) between them -- an item cost two to three dependent memory round trips instead of one
    // (the strip loop ran at ~6 k clocks per strip, tools/gpu_wgrad_trace.py).
    float4 stg[8];
    unsigned ok = 0;                                  // bit e: element e of the staged item is inside the image
    auto issue = [&](int st) {
        const int img = st / SPI, y0 = (st % SPI) * RS;
        const int gy = y0 - 1 + sr, gyc = gy < 0 ? 0 : gy >= H ? H - 1 : gy;
        const int c = 4 * cq + (isx ? 32 * nb0 : 32 * mb);          // channel in the tensor
        // (threads without an item read like a dY item of row 0: always a valid address)
        const float* rowp = !isx ? a.dY + ((size_t)(img * H + y0 + (isy ? sr : 0)) * W) * COUT + c
                                 : c < C0 ? a.in0 + ((size_t)(img * H + gyc) * W) * C0 + c
                                          : a.in1 + ((size_t)(img * H + gyc) * W) * C1 + (c - C0);
        const int ps = !isx ? COUT : c < C0 ? C0 : C1;
        const bool row_ok = isy || (isx && gy >= 0 && gy < H);
        ok = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int x = 8 * sg + e, xc = x < W ? x : W - 1;
            stg[e] = *reinterpret_cast<const float4*>(rowp + (size_t)xc * ps);
            ok |= (row_ok && x < W) ? 1u << e : 0u;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const __bf16* ya = Yt + (size_t)n * YCS;                           // this lane's dY channel (of block mb)
    const __bf16* xb = Xt + (size_t)(nbl * 32 + n) * XCS + 8;          // this lane's X channel, col of x = 0

    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int st = blockIdx.x;
    if (st < nstrips) issue(st);
    for (; st < nstrips; st += gridDim.x) {
        __syncthreads();                              // everyone is done reading the previous strip (and the zero fill)
        const unsigned okc = ok;
        if (isx || isy) {
            __bf16* dst = isx ? Xt + (size_t)(4 * cq) * XCS + sr * PX + 8 + 8 * sg : Yt + (size_t)(4 * cq) * YCS + sr * PY + 8 * sg;
            const int cstride = isx ? XCS : YCS;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float v8[8] = {ch == 0 ? stg[0].x : ch == 1 ? stg[0].y : ch == 2 ? stg[0].z : stg[0].w,
                                     ch == 0 ? stg[1].x : ch == 1 ? stg[1].y : ch == 2 ? stg[1].z : stg[1].w,
                                     ch == 0 ? stg[2].x : ch == 1 ? stg[2].y : ch == 2 ? stg[2].z : stg[2].w,
                                     ch == 0 ? stg[3].x : ch == 1 ? stg[3].y : ch == 2 ? stg[3].z : stg[3].w,
                                     ch == 0 ? stg[4].x : ch == 1 ? stg[4].y : ch == 2 ? stg[4].z : stg[4].w,
                                     ch == 0 ? stg[5].x : ch == 1 ? stg[5].y : ch == 2 ? stg[5].z : stg[5].w,
                                     ch == 0 ? stg[6].x : ch == 1 ? stg[6].y : ch == 2 ? stg[6].z : stg[6].w,
                                     ch == 0 ? stg[7].x : ch == 1 ? stg[7].y : ch == 2 ? stg[7].z : stg[7].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (okc >> e & 1u) ? v8[e] : 0.f;
                wbf8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v8[e];
                *reinterpret_cast<wbf8*>(dst + (size_t)ch * cstride) = o;
                if (isy && nb0 == 0)                  // bias gradient: column sums of dY (this block's 32 channels), from the fp32 values
                    bsum[ch] += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
            }
        }
        __syncthreads();
        if (st + (int)gridDim.x < nstrips) issue(st + gridDim.x);      // flies under this strip's MFMAs
        for (int j = ks; j < NCH; j += KS) {
            const int gi = 2 * j + hi, r = gi / G, xg = gi % G;
            const wbf8 A = *reinterpret_cast<const wbf8*>(ya + r * PY + 8 * xg);
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const __bf16* row = xb + (r + ty) * PX + 8 * xg;                 // X row y + ty - 1, col of the group's x0
                const uint4 mid = *reinterpret_cast<const uint4*>(row);
                const unsigned wp = *reinterpret_cast<const unsigned*>(row - 2);  // elements x0-2, x0-1
                const unsigned wn = *reinterpret_cast<const unsigned*>(row + 8);  // elements x0+8, x0+9
                const uint4 left = {__builtin_amdgcn_alignbyte(mid.x, wp, 2), __builtin_amdgcn_alignbyte(mid.y, mid.x, 2),
                                    __builtin_amdgcn_alignbyte(mid.z, mid.y, 2), __builtin_amdgcn_alignbyte(mid.w, mid.z, 2)};
                const uint4 right = {__builtin_amdgcn_alignbyte(mid.y, mid.x, 2), __builtin_amdgcn_alignbyte(mid.z, mid.y, 2),
                                     __builtin_amdgcn_alignbyte(mid.w, mid.z, 2), __builtin_amdgcn_alignbyte(wn, mid.w, 2)};
                acc[3 * ty + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(wbf8, left), acc[3 * ty + 0], 0, 0, 0);
                acc[3 * ty + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(wbf8, mid), acc[3 * ty + 1], 0, 0, 0);
                acc[3 * ty + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(wbf8, right), acc[3 * ty + 2], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: as conv3_wgrad_kernel (the D layout of the bf16 MFMA is the same 32x32 map) ----------------------
    if (nb0 == 0) {
        __syncthreads();
        // per-channel totals: threads of the dY items hold 4 channels each for their (row, group)
        float* red = wg_lds;                                          // [32] after zeroing
        for (int c = tid; c < COUTW; c += 512) red[c] = 0.f;
        __syncthreads();
        if (isy) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) atomicAdd(red + 4 * cq + ch, bsum[ch]);      // LDS, <= RS*G addends per channel
        }
        __syncthreads();
        if (tid < COUTW)
            a.partial[(size_t)gridDim.y * gridDim.x * BPG * 9 * 1024 + (size_t)blockIdx.x * COUT + 32 * mb + tid] = red[tid];
    }
    float4* slot = reinterpret_cast<float4*>(wg_lds);                 // [wave][tap of the round][r4][lane]
    constexpr int RN = 9 * 1024;
    float* part = a.partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BPG) * RN;
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < 3; ++tl)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                slot[((wave * 3 + tl) * 4 + r4) * 64 + lane] =
                    make_float4(acc[3 * g3 + tl][4 * r4], acc[3 * g3 + tl][4 * r4 + 1], acc[3 * g3 + tl][4 * r4 + 2],
                                acc[3 * g3 + tl][4 * r4 + 3]);
        __syncthreads();
        for (int v = tid; v < BPG * 3 * 4 * 64; v += 512) {
            const int ln = v & 63, r4 = (v >> 6) & 3, tl = (v >> 8) % 3, b = v / 768;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const float4 x = slot[(((b * KS + q) * 3 + tl) * 4 + r4) * 64 + ln];
                sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
            }
            const int nn = ln & 31, m0 = 8 * r4 + 4 * (ln >> 5);      // D rows of registers 4*r4 .. 4*r4+3: m0 .. m0+3
            float* dst = part + ((size_t)(b * 9 + 3 * g3 + tl) * 32 + m0) * 32 + nn;
            dst[0] = sum.x; dst[32] = sum.y; dst[64] = sum.z; dst[96] = sum.w;
        }
    }
}

// ---- the same weight gradient for layers with at most four 32x32 blocks: the NINE TAPS go to nine waves ----------------------
// In conv3_wgrad_bf16_kernel a wave owns a block for all nine taps (144 accumulator registers) and the waves of a block split the
// pixels.  For the layers with one to four blocks (32 -> 32 and 32 + 32 -> 32 at 40x40, 32 -> 64 and 64 -> 64 at 20x20) that leaves
//   * no registers for a second strip of loads in flight: a strip is ~600 clocks of MFMAs against a ~6 k-clock memory round trip,
//     one strip deep the loop runs at one strip per round trip (tools/gpu_wgrad_trace.py, profiles/r03/wgrad3_bf16_trace_layer0.txt);
//   * an epilogue of three LDS rounds in which the 8 waves' tiles are summed (9.6 k of the launch's 44 k clocks).
// Here wave t takes tap t of EVERY block and every k-chunk of the strip: 16 x NBLK accumulator registers, the same number of
// MFMAs per wave, no cross-wave sum at all (a wave writes its tiles straight to the partial image), and two strips of loads in
// flight (ping-pong staging registers).  Workgroup barriers inside the loop are s_barrier after lgkmcnt(0) only: __syncthreads()
// also waits for vmcnt(0), i.e. for the prefetched loads.  Staging, partial-image layout and reduce kernel are unchanged.
template <int C0, int C1, int COUT, int H, int RS>
__global__ __launch_bounds__(576) void conv3_wgrad_bf16_taps_kernel(Wgrad3Args a) {
    constexpr int W = H, CIN = C0 + C1;
    constexpr int NBK = CIN / 32, NMB = COUT / 32, NBLK = NMB * NBK;
    static_assert(NBLK <= 4, "one tap per wave: few blocks");
    constexpr bool DEEP = NBLK == 1;                  // two strips of loads in flight; with 2-4 blocks (32-64 accumulator registers) the second
                                                      // staging set no longer fits the 168 registers of a 9-wave workgroup (100-400 B of scratch)
    constexpr int G = (W + 7) / 8;
    constexpr int PY = 8 * G, PX = 8 * G + 16;
    constexpr int XR = RS + 2;
    constexpr int NVX = (CIN / 4) * XR * G, NVY = (COUT / 4) * RS * G, NV = NVX + NVY;
    static_assert(NV <= 512, "one staging item per thread");
    constexpr int XCS = wg_cs(XR * PX), YCS = wg_cs(RS * PY);        // channel strides (see wg_cs)
    constexpr int XELEMS = CIN * XCS;
    constexpr int NCH = RS * G / 2;
    static_assert((RS * G) % 2 == 0 && H % RS == 0, "strip geometry");
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    __bf16* Xt = reinterpret_cast<__bf16*>(wg_lds);
    __bf16* Yt = Xt + XELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = tap
    const int n = lane & 31, hi = lane >> 5;
    const int ty = wave / 3, tx = wave - 3 * ty;

    for (int v = tid; v < XELEMS / 8; v += 576) reinterpret_cast<uint4*>(Xt)[v] = make_uint4(0, 0, 0, 0);

    const bool isx = tid < NVX, isy = !isx && tid < NV;
    const int u = isx ? tid : tid - NVX;
    const int cq = isx ? u % (CIN / 4) : u % (COUT / 4);
    const int rest = isx ? u / (CIN / 4) : u / (COUT / 4);
    const int sg = rest % G, sr = rest / G;
    constexpr int SPI = H / RS;
    const int nstrips = a.nimg * SPI;
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    float4 stgA[8], stgB[8];
    unsigned okA = 0, okB = 0;                        // bit e: element e of the staged item is inside the image
    // (unconditional loads from clamped addresses, zeroed when stored: see conv3_wgrad_bf16_kernel)
    auto issue = [&](int st, float4 (&stg)[8], unsigned& ok) {
        const int img = st / SPI, y0 = (st % SPI) * RS;
        const int gy = y0 - 1 + sr, gyc = gy < 0 ? 0 : gy >= H ? H - 1 : gy, c = 4 * cq;
        // (threads without an item read like a dY item of row 0: always a valid address)
        const float* rowp = !isx ? a.dY + ((size_t)(img * H + y0 + (isy ? sr : 0)) * W) * COUT + c
                                 : c < C0 ? a.in0 + ((size_t)(img * H + gyc) * W) * C0 + c
                                          : a.in1 + ((size_t)(img * H + gyc) * W) * C1 + (c - C0);
        const int ps = !isx ? COUT : c < C0 ? C0 : C1;
        const bool row_ok = isy || (isx && gy >= 0 && gy < H);
        ok = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int x = 8 * sg + e, xc = x < W ? x : W - 1;
            stg[e] = *reinterpret_cast<const float4*>(rowp + (size_t)xc * ps);
            ok |= (row_ok && x < W) ? 1u << e : 0u;
        }
    };

    f32x16 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int st = blockIdx.x;
    const int gstep = (int)gridDim.x;
    if (st < nstrips) issue(st, stgA, okA);
    if (DEEP && st + gstep < nstrips) issue(st + gstep, stgB, okB);
    __syncthreads();                                  // the zero fill of the X image
    auto strip = [&](float4 (&stg)[8], unsigned& ok, int st_) {
        lds_barrier();                                // everyone is done reading the previous strip
        const unsigned okc = ok;
        if (isx || isy) {
            __bf16* dst = isx ? Xt + (size_t)(4 * cq) * XCS + sr * PX + 8 + 8 * sg : Yt + (size_t)(4 * cq) * YCS + sr * PY + 8 * sg;
            const int cstride = isx ? XCS : YCS;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float v8[8] = {ch == 0 ? stg[0].x : ch == 1 ? stg[0].y : ch == 2 ? stg[0].z : stg[0].w,
                                     ch == 0 ? stg[1].x : ch == 1 ? stg[1].y : ch == 2 ? stg[1].z : stg[1].w,
                                     ch == 0 ? stg[2].x : ch == 1 ? stg[2].y : ch == 2 ? stg[2].z : stg[2].w,
                                     ch == 0 ? stg[3].x : ch == 1 ? stg[3].y : ch == 2 ? stg[3].z : stg[3].w,
                                     ch == 0 ? stg[4].x : ch == 1 ? stg[4].y : ch == 2 ? stg[4].z : stg[4].w,
                                     ch == 0 ? stg[5].x : ch == 1 ? stg[5].y : ch == 2 ? stg[5].z : stg[5].w,
                                     ch == 0 ? stg[6].x : ch == 1 ? stg[6].y : ch == 2 ? stg[6].z : stg[6].w,
                                     ch == 0 ? stg[7].x : ch == 1 ? stg[7].y : ch == 2 ? stg[7].z : stg[7].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (okc >> e & 1u) ? v8[e] : 0.f;
                wbf8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v8[e];
                *reinterpret_cast<wbf8*>(dst + (size_t)ch * cstride) = o;
                if (isy)                              // bias gradient: column sums of dY, from the fp32 values
                    bsum[ch] += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
            }
        }
        if (st_ + (DEEP ? 2 : 1) * gstep < nstrips) issue(st_ + (DEEP ? 2 : 1) * gstep, stg, ok);   // into the registers just drained
        lds_barrier();
#pragma unroll 2
        for (int j = 0; j < NCH; ++j) {
            const int gi = 2 * j + hi, r = gi / G, xg = gi % G;
            wbf8 A[NMB], B[NBK];
#pragma unroll
            for (int m = 0; m < NMB; ++m) A[m] = *reinterpret_cast<const wbf8*>(Yt + (size_t)(m * 32 + n) * YCS + r * PY + 8 * xg);
#pragma unroll
            for (int k = 0; k < NBK; ++k) {
                const __bf16* row = Xt + (size_t)(k * 32 + n) * XCS + 8 + (r + ty) * PX + 8 * xg;   // X row y + ty - 1, col of the group's x0
                const uint4 mid = *reinterpret_cast<const uint4*>(row);
                uint4 sel = mid;
                if (tx == 0) {
                    const unsigned wp = *reinterpret_cast<const unsigned*>(row - 2);                  // elements x0-2, x0-1
                    sel = uint4{__builtin_amdgcn_alignbyte(mid.x, wp, 2), __builtin_amdgcn_alignbyte(mid.y, mid.x, 2),
                                __builtin_amdgcn_alignbyte(mid.z, mid.y, 2), __builtin_amdgcn_alignbyte(mid.w, mid.z, 2)};
                } else if (tx == 2) {
                    const unsigned wn = *reinterpret_cast<const unsigned*>(row + 8);                  // elements x0+8, x0+9
                    sel = uint4{__builtin_amdgcn_alignbyte(mid.y, mid.x, 2), __builtin_amdgcn_alignbyte(mid.z, mid.y, 2),
                                __builtin_amdgcn_alignbyte(mid.w, mid.z, 2), __builtin_amdgcn_alignbyte(wn, mid.w, 2)};
                }
                B[k] = __builtin_bit_cast(wbf8, sel);
            }
#pragma unroll
            for (int m = 0; m < NMB; ++m)
#pragma unroll
                for (int k = 0; k < NBK; ++k)
                    acc[m * NBK + k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[k], acc[m * NBK + k], 0, 0, 0);
        }
    };
    if constexpr (DEEP) {
        for (;;) {
            if (st >= nstrips) break;
            strip(stgA, okA, st); st += gstep;
            if (st >= nstrips) break;
            strip(stgB, okB, st); st += gstep;
        }
    } else {
        for (; st < nstrips; st += gstep) strip(stgA, okA, st);
    }
    // ---- epilogue: every wave writes its tap's tiles (D layout: lane (n, hi), register r <-> row (r&3) + 8(r>>2) + 4hi) ------------
    constexpr int RN = 9 * 1024;
    float* part = a.partial + (size_t)blockIdx.x * NBLK * RN;
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part[((size_t)(b * 9 + wave) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + n] = acc[b][r];
    // per-channel totals of dY: threads of the dY items hold 4 channels each for their (row, group)
    __syncthreads();
    float* red = wg_lds;
    for (int c = tid; c < COUT; c += 576) red[c] = 0.f;
    __syncthreads();
    if (isy) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) atomicAdd(red + 4 * cq + ch, bsum[ch]);      // LDS, <= RS*G addends per channel
    }
    __syncthreads();
    if (tid < COUT) a.partial[(size_t)gridDim.x * NBLK * RN + (size_t)blockIdx.x * COUT + tid] = red[tid];
}

template <int C0, int C1, int COUT, int H, int RS>
static int launch_wgrad3_bf16(const Wgrad3Args& a, hipStream_t s, Wg3RedArgs* defer = nullptr) {
    constexpr int CIN = C0 + C1, W = H;
    constexpr int NBLK = (COUT / 32) * (CIN / 32), BPG = NBLK < 8 ? NBLK : 8, NY = NBLK / BPG;
    constexpr int G = (W + 7) / 8;
    constexpr size_t strip = ((size_t)CIN * wg_cs((RS + 2) * (8 * G + 16)) + (size_t)COUT * wg_cs(RS * 8 * G)) * 2;
    constexpr size_t lds = strip > 8 * 3 * 4 * 64 * 16 ? strip : 8 * 3 * 4 * 64 * 16;      // strip or the epilogue slots (96 KiB)
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int nstrips = a.nimg * (H / RS);
    int gx = 256 / NY;
    if (gx > nstrips) gx = nstrips;
    if constexpr (NBLK <= 4) {                        // one tap per wave (conv3_wgrad_bf16_taps_kernel)
        static const bool taps = [] { const char* e = getenv("GIGA_WGRAD_TAPS"); return !e || atoi(e) != 0; }();
        if (taps) {
            auto kern = conv3_wgrad_bf16_taps_kernel<C0, C1, COUT, H, RS>;
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)strip);
            GIGA_LAUNCH(kern, dim3(gx), dim3(576), strip, s, a);
            if (defer) { wg3_defer(defer, a.partial, gx, NY, BPG, CIN, COUT, a.dW, a.db); return hipGetLastError() == hipSuccess ? 0 : -10; }
            GIGA_LAUNCH((wgrad3_reduce_kernel<CIN, BPG>), dim3(9 * 1024 / 64 + 1, NBLK), dim3(256), 0, s, a.partial, gx,
                               a.dW, a.db, COUT, NY);
            return hipGetLastError() == hipSuccess ? 0 : -10;
        }
    }
    if constexpr (NBLK > 4) {                         // the output split over the workgroups too (conv3_wgrad_bf16_ns_kernel)
        static const bool nsplit = [] { const char* e = getenv("GIGA_WGRAD_NSPLIT"); return !e || atoi(e) != 0; }();
        if (nsplit) {
            constexpr int BX = 2, NYS = NBLK / BX;
            constexpr size_t strip_ns = ((size_t)32 * BX * wg_cs((RS + 2) * (8 * G + 16)) + (size_t)32 * wg_cs(RS * 8 * G)) * 2;
            constexpr size_t slots = (size_t)8 * 3 * 4 * 64 * 16;                          // the epilogue's LDS slots (96 KiB)
            constexpr size_t lds_ns = strip_ns > slots ? strip_ns : slots;
            int gxs = 256 / NYS;
            if (gxs > nstrips) gxs = nstrips;
            auto kns = conv3_wgrad_bf16_ns_kernel<C0, C1, COUT, H, RS, BX>;
            giga::dyn_lds_once(reinterpret_cast<const void*>(kns), (int)lds_ns);
            GIGA_LAUNCH(kns, dim3(gxs, NYS), dim3(512), lds_ns, s, a);
            if (defer) { wg3_defer(defer, a.partial, gxs, NYS, BX, CIN, COUT, a.dW, a.db); return hipGetLastError() == hipSuccess ? 0 : -10; }
            GIGA_LAUNCH((wgrad3_reduce_kernel<CIN, BX>), dim3(9 * 1024 / 64 + 1, NBLK), dim3(256), 0, s, a.partial, gxs,
                               a.dW, a.db, COUT, NYS);
            return hipGetLastError() == hipSuccess ? 0 : -10;
        }
    }
    auto kern = conv3_wgrad_bf16_kernel<C0, C1, COUT, H, RS>;
    giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
    GIGA_LAUNCH(kern, dim3(gx, NY), dim3(512), lds, s, a);
    if (defer) { wg3_defer(defer, a.partial, gx, NY, BPG, CIN, COUT, a.dW, a.db); return hipGetLastError() == hipSuccess ? 0 : -10; }
    GIGA_LAUNCH((wgrad3_reduce_kernel<CIN, BPG>), dim3(9 * 1024 / 64 + 1, NBLK), dim3(256), 0, s, a.partial, gx,
                       a.dW, a.db, COUT, NY);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

// ------------------------------- conv_in backward ----------------------------------------------------------
// Same decomposition as convin_project_kernel: workgroup = (x-part, iy-group x channel half, scene), the haloed TSDF
// sub-volume is staged in LDS once, every wave owns SXW slices and runs without workgroup barriers; units of 16 voxels x
// 16 channels.  Per unit: recompute pre = conv_in(x) + b (7 MFMAs), dF = (pre > 0) * (gxz + gxy + gyz) / 40, then
// dW^T[tap][ch] += X^T[tap][voxel] * dF[voxel][ch] as 2 tap halves x 4 k-steps = 8 more MFMAs (the D registers of the
// forward unit ARE the B operand of these).  The yz-plane gradient of a lane's 100 voxels does not depend on the slice and
// lives in registers; the xz / xy rows of a slice are loaded into registers at the top of the slice.  Every workgroup
// leaves one partial (27x16 weights + 16 biases, fixed-order sum of its 8 waves through LDS); convin_bwd_reduce_kernel adds
// the workgroups up (no same-address atomics: those are resolved outside the XCD-local L2, one fabric operation each).
constexpr int CB_RS = 44, CB_ROWS = 12;           // staged sub-volume: (XW+2) x 12 rows x 44 floats (iz + 1 in [0, 41])
constexpr int CB_PART = 27 * 16 + 16;             // floats per workgroup partial: dW^T[tap][ch16], db[ch16]
constexpr size_t cb_lds_bytes(int sxw) { return (size_t)(8 * sxw + 2) * CB_ROWS * CB_RS * sizeof(float); }

// MASK: the forward (convin_project_kernel<.., MASK = true>, GIGA_CONVIN_MASK) has left the sign bits of the pre-activations -- per
// (scene, workgroup row, ix) and lane one 16-byte word, value k = (zg * 5 + ip) * 4 + r in word k >> 5 at bit (n - 1 - (k & 31)),
// n = 32 (4 in the last word) -- and the seven recomputation MFMAs of every unit fall away (8 instead of 15 per unit).
template <int SXW, bool MASK = false>
__global__ __launch_bounds__(512) void convin_bwd_kernel(const float* __restrict__ tsdf, const float* __restrict__ wpk,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ gplanes,   // [3][B][40][40][32]
                                                         float* __restrict__ partial,         // [workgroup][CB_PART]
                                                         int B, const uint4* __restrict__ relu_mask = nullptr) {
    constexpr int XW = 8 * SXW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int xp = blockIdx.x, grp = blockIdx.y >> 1, chh = blockIdx.y & 1, b = blockIdx.z;
    const int x0 = xp * XW;
    const float* vol = tsdf + (size_t)b * RES * RES * RES;
    for (int v = tid; v < (XW + 2) * CB_ROWS * 10; v += 512) {
        const int q = v % 10, row = v / 10, yl = row % CB_ROWS, xl = row / CB_ROWS;
        const int ix = x0 - 1 + xl, iy = 10 * grp - 1 + yl;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ix >= 0 && ix < RES && iy >= 0 && iy < RES)
            val = *reinterpret_cast<const float4*>(vol + ((size_t)ix * RES + iy) * RES + 4 * q);
        float* dst = lds + row * CB_RS + 1 + 4 * q;
        dst[0] = val.x; dst[1] = val.y; dst[2] = val.z; dst[3] = val.w;
        if (q == 0) dst[-1] = 0.f;
        if (q == 9) dst[4] = 0.f;
    }
    float wreg[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) wreg[s] = wpk[(chh * 7 + s) * 64 + lane];
    const int ch = 16 * chh + j;
    const float bn = bias[ch];
    const size_t img_stride = (size_t)RES * RES * CD;
    const float* gxz = gplanes + ((size_t)0 * B + b) * img_stride;
    const float* gxy = gplanes + ((size_t)1 * B + b) * img_stride;
    const float* gyz = gplanes + ((size_t)2 * B + b) * img_stride;
    const float inv = 1.0f / RES;
    // forward A operand: row i = lane&15 is voxel (iy_l = i>>3, iz_l = 4*((i>>2)&1) + (i&3)); k-slot g -> tap 4s+g
    int abase[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        int t = 4 * s + g;
        t = t > 26 ? 26 : t;
        abase[s] = ((t / 9) * CB_ROWS + (j >> 3) + (t / 3) % 3) * CB_RS + 4 * ((j >> 2) & 1) + (j & 3) + t % 3;
    }
    // backward A operand: row = tap (lane&15 + 16*th), k-slot g -> voxel row 4g + r of the unit (iy_l = g>>1, iz_l = 4(g&1)+r)
    int tbase[2];
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        int t = j + 16 * th;
        t = t > 26 ? 26 : t;                      // rows >= 27 of dW^T are never written back
        tbase[th] = ((t / 9) * CB_ROWS + (g >> 1) + (t / 3) % 3) * CB_RS + 4 * (g & 1) + t % 3;
    }
    // the yz-plane gradient of this lane's 100 voxels does not depend on the slice: registers, loaded once
    f32x4v gz[5][5];
#pragma unroll
    for (int ip = 0; ip < 5; ++ip)
#pragma unroll
        for (int zg = 0; zg < 5; ++zg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gz[ip][zg][r] = gyz[((8 * zg + 4 * (g & 1) + r) * RES + grp * 10 + 2 * ip + (g >> 1)) * CD + ch];
    f32x4v accw[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
    float accb = 0.f;
    const f32x4v bias4 = {bn, bn, bn, bn};
    __syncthreads();

    for (int sx = 0; sx < SXW; ++sx) {
        const int ixl = wave * SXW + sx, ix = x0 + ixl;
        const int so = ixl * CB_ROWS * CB_RS;
        // upstream gradients of this slice that do not depend on the unit: xy row per iy-pair, xz rows per (zg, r)
        float gy[5];
        f32x4v gx[5];
        unsigned mwv[4] = {0u, 0u, 0u, 0u};
        if constexpr (MASK) {
            const uint4 m4 = relu_mask[(((size_t)b * 8 + blockIdx.y) * RES + ix) * 64 + lane];
            mwv[0] = m4.x; mwv[1] = m4.y; mwv[2] = m4.z; mwv[3] = m4.w;
        }
#pragma unroll
        for (int ip = 0; ip < 5; ++ip) gy[ip] = gxy[((grp * 10 + 2 * ip + (g >> 1)) * RES + ix) * CD + ch];
#pragma unroll
        for (int zg = 0; zg < 5; ++zg)
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[zg][r] = gxz[((8 * zg + 4 * (g & 1) + r) * RES + ix) * CD + ch];
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) {
            // forward recompute: five independent 7-MFMA chains (bias in the C operand, as the forward does) -- or the stored sign bits
            f32x4v d[5];
            if constexpr (!MASK) {
#pragma unroll
                for (int ip = 0; ip < 5; ++ip) d[ip] = bias4;
#pragma unroll
                for (int s = 0; s < 7; ++s)
#pragma unroll
                    for (int ip = 0; ip < 5; ++ip)
                        d[ip] = mfma32_16(lds[abase[s] + so + 2 * ip * CB_RS + 8 * zg], wreg[s], d[ip]);
            }
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) {
                const int uo = so + 2 * ip * CB_RS + 8 * zg;
                f32x4v dF;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gsum = gx[zg][r] + gy[ip] + gz[ip][zg][r];
                    bool pos;
                    if constexpr (MASK) {
                        const int k = (zg * 5 + ip) * 4 + r, wd = k >> 5, bit = (wd == 3 ? 3 : 31) - (k & 31);
                        pos = ((mwv[wd] >> bit) & 1u) == 0u;                 // sign bit clear
                    } else {
                        pos = d[ip][r] > 0.f;
                    }
                    dF[r] = pos ? gsum * inv : 0.f;
                    accb += dF[r];
                }
#pragma unroll
                for (int th = 0; th < 2; ++th)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accw[th] = mfma32_16(lds[tbase[th] + uo + r], dF[r], accw[th]);
            }
        }
    }
    // ---- workgroup partial: fixed-order sum of the 8 waves.  accw[th][r]: dW^T[tap = 16th + 4g + r][ch] ----------------
    accb += __shfl_xor(accb, 16);
    accb += __shfl_xor(accb, 32);
    __syncthreads();                                  // every wave is done with the sub-volume
    float* red = lds;                                 // [wave][CB_PART]
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tap = 16 * th + 4 * g + r;
            if (tap < 27) red[wave * CB_PART + tap * 16 + j] = accw[th][r];
        }
    if (g == 0) red[wave * CB_PART + 27 * 16 + j] = accb;
    __syncthreads();
    if (tid < CB_PART) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w * CB_PART + tid];
        const int wgi = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[(size_t)wgi * CB_PART + tid] = sum;
    }
}

// dW[ch][tap] / db[ch] = sum over the scene's workgroups of one (channel half): partial[scene][grp*2+chh][xp][CB_PART]
__global__ __launch_bounds__(256) void convin_bwd_reduce_kernel(const float* __restrict__ partial, int B, int nxp,
                                                                float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float sm[256];
    const int e = blockIdx.x, chh = blockIdx.y;       // e: element of the partial (tap*16 + j, or 432 + j)
    float s = 0.f;
    const int per_scene = 8 * nxp;                    // workgroups per scene, (grp*2+chh) major, x-part minor
    for (int i = threadIdx.x; i < B * 4 * nxp; i += 256) {
        const int bq = i / (4 * nxp), rest = i % (4 * nxp), grp = rest / nxp, xp = rest % nxp;
        s += partial[((size_t)bq * per_scene + (grp * 2 + chh) * nxp + xp) * CB_PART + e];
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int jj = e & 15, ch = 16 * chh + jj;
        if (e < 27 * 16) dW[ch * 27 + (e >> 4)] += sm[0];
        else db[ch] += sm[0];
    }
}

// ------------------------------- the library's side stream (giga_side.h) -------------------------------------------
static std::mutex g_side_mutex[64];               // one per device: callers on different devices do not wait for each other
static SideStream g_side[64];
SideScope::SideScope(hipStream_t main, bool enable) : main_(main) {
    if (!enable) return;
    // the device that OWNS the caller's stream (the null stream belongs to the current device); a caller whose current device is
    // another one gets the inactive scope: recording this device's events on a foreign stream would fail halfway through the pass
    int dev = 0, sdev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    if (main && (hipStreamGetDevice(main, &sdev) != hipSuccess || sdev != dev)) return;
    g_side_mutex[dev].lock();
    SideStream* w = &g_side[dev];
    if (!w->ok) {
        bool ok = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 16 && ok; ++i) ok = hipEventCreateWithFlags(&w->fork[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&w->join[i], hipEventDisableTiming) == hipSuccess;
        w->ok = ok;
    }
    if (w->ok) { side_ = w; dev_ = dev; }
    else g_side_mutex[dev].unlock();
}
SideScope::~SideScope() {
    if (!side_) return;
    if (nfork_ > 0 && njoin_ == 0) (void)join();      // an error path left the side stream forked (inside a capture: unjoined): join it anyway
    g_side_mutex[dev_].unlock();
}
int SideScope::fork() {
    if (!side_) return 0;
    hipEvent_t e = side_->fork[nfork_++ & 15];
    return (hipEventRecord(e, main_) == hipSuccess && hipStreamWaitEvent(side_->stream, e, 0) == hipSuccess) ? 0 : -10;
}
int SideScope::join() {
    if (!side_) return 0;
    hipEvent_t e = side_->join[njoin_++ & 3];
    return (hipEventRecord(e, side_->stream) == hipSuccess && hipStreamWaitEvent(main_, e, 0) == hipSuccess) ? 0 : -10;
}
void side_streams_forget() {
    for (int d = 0; d < 64; ++d) {
        std::lock_guard<std::mutex> lk(g_side_mutex[d]);
        g_side[d] = SideStream{};
    }
}

// ------------------------------- driver -----------------------------------------------------------------------
// forward activations: the encoder workspace (giga_encoder.hip::EncWs, fp32).  Gradient workspace carve:
// (struct BwdWs: giga_bwd_mega.h)
BwdWs enc_bwd_workspace(int B) {
    const size_t n = 3 * (size_t)B;
    BwdWs w{};
    size_t at = 0;
    auto take = [&](size_t elems) { size_t o = at; at += align_up(elems * 4, 256); return o; };
    w.gA6 = take(n * 1600 * 32); w.gA5 = take(n * 1600 * 32); w.gC1 = take(n * 1600 * 64);
    w.gA4 = take(n * 400 * 64);  w.gA3 = take(n * 400 * 64);  w.gC0 = take(n * 400 * 128);
    w.gS2 = take(n * 100 * 128); w.gA2 = take(n * 100 * 128); w.gQ1 = take(n * 100 * 64);
    w.gS1 = take(n * 400 * 64);  w.gA1 = take(n * 400 * 64);  w.gQ0 = take(n * 400 * 32);
    w.gS0 = take(n * 1600 * 32); w.gA0 = take(n * 1600 * 32); w.gP0 = take(n * 1600 * 32);
    w.WG = take((size_t)WG3_MAX_PARTS * 32 * 288 + WG3_BIAS_FLOATS);      // per-workgroup weight-gradient partials (conv3_wgrad_kernel)
    // one region per 3x3 layer for the single reduce launch at the end (wgrad3_reduce_all_kernel): 256 x min(blocks, 8) partial
    // images of 36 KiB + the bias partials; 368 MB in all, whatever the batch size
    for (int l = 0; l < 12; ++l) {
        w.WG3[l] = 0;
        if (kConv[l].kind != CONV3) continue;
        const int nblk = (kConv[l].cout / 32) * ((kConv[l].cin0 + kConv[l].cin1) / 32);
        w.WG3[l] = take((size_t)256 * (nblk < 8 ? nblk : 8) * 32 * 288 + WG3_BIAS_FLOATS);
    }
    w.CINP = take((size_t)B * 8 * 5 * CB_PART);    // conv_in's workgroup partials (their own: the backward runs beside the weight gradients' stream)
    w.total = at;                                  // (+ MEGA_SYNC_WORDS words behind it: the persistent data-gradient kernel's counters)
    return w;
}

EncWs enc_workspace(int B, int precision);
int enc_nxp(int B);

// MATH: arithmetic of the thirteen data-gradient convolutions (MATH_NATIVE fp32 MFMA, or MATH_BF16: bf16 operands from the
// backward blob's bf16 images, fp32 accumulate and fp32 gradients in memory); everything else is fp32.
template <int V> using IC = std::integral_constant<int, V>;
template <int MATH>
static int encoder_backward_impl(const float* tsdf, const uint8_t* blob, const uint8_t* bwd_blob, const uint8_t* fws,
                                 float* gplanes /* [3B][40][40][32], in: dLoss/dPlanes, clobbered */, uint8_t* gws,
                                 float* grads /* flat, state-dict order */, int head_present, int B, hipStream_t s, bool convin_mask,
                                 SideScope& side) {
    if (B <= 0) return 0;
    const PackOff ko = pack_offsets();
    const BwdPackOff bo = bwd_pack_offsets();
    const ParamOff po = param_offsets(head_present);
    const EncWs f = enc_workspace(B, 0);
    const BwdWs g = enc_bwd_workspace(B);
    const int nimg = 3 * B;
    auto F = [&](size_t off) { return reinterpret_cast<const float*>(fws + off); };
    auto G = [&](size_t off) { return reinterpret_cast<float*>(gws + off); };
    int rc = 0;
    // the ten reduces of the 3x3 layers' partial weight gradients as ONE launch behind the last of them (GIGA_WGRAD_ONE_REDUCE=0: one
    // launch per layer, right behind its weight-gradient kernel, as before round 5)
    const bool one_reduce = [] { const char* e = getenv("GIGA_WGRAD_ONE_REDUCE"); return !e || atoi(e) != 0; }();     // (per call)
    Wg3RedArgs RED{};
    // the weight gradients on the device's side stream (`side`: giga_side.h; inactive = on the caller's stream, between the data gradients)
    const hipStream_t ws = side.stream();
    auto fork = [&]() { rc |= side.fork(); };
    // weight gradient of layer l: R = dPre (channels = cout), columns = layer input (in0 [, in1])
    auto wgrad3_now = [&](int l, const float* dpre, const float* in0, const float* in1, int H) {
        const ConvLayerDesc& d = kConv[l];
        const int cin = d.cin0 + d.cin1, taps = d.kind == CONV3 ? 9 : 1;
        WgradArgs a{};
        a.R = dpre; a.csR = d.cout; a.coR = 0;
        a.C0p = in0; a.C1p = in1; a.csC0 = d.cin0; a.csC1 = d.cin1; a.nC0 = d.cin0;
        a.dW = grads + po.conv_w[l]; a.sM = cin * taps; a.sN = taps; a.sT = 1;
        a.kind = d.kind; a.taps = taps; a.Mb = d.cout / 32; a.Nb = cin / 32;
        a.nimg = nimg; a.H = H; a.W = H;
        a.partial = G(g.WG);
        a.db = grads + po.conv_b[l]; a.nbias = d.cout;          // bias gradient = column sums of dPre, folded into the same two launches
        if (d.kind == CONV3) {
            Wgrad3Args w3{dpre, in0, in1, grads + po.conv_w[l], grads + po.conv_b[l], one_reduce ? G(g.WG3[l]) : G(g.WG), nimg};
            Wg3RedArgs* defer = one_reduce ? &RED : nullptr;
#define WG3(...) (MATH == MATH_BF16 ? launch_wgrad3_bf16<__VA_ARGS__>(w3, ws, defer) : launch_wgrad3<__VA_ARGS__>(w3, ws, defer))
            switch (l) {   // <C0, C1, COUT, H, rows per strip>
                case 0: case 1: case 11: rc |= WG3(32, 0, 32, 40, 4); break;
                case 10: rc |= WG3(32, 32, 32, 40, 2); break;
                case 2: rc |= WG3(32, 0, 64, 20, 4); break;
                case 3: case 8: rc |= WG3(64, 0, 64, 20, 4); break;
                case 7: rc |= WG3(64, 64, 64, 20, 2); break;
                case 4: rc |= WG3(64, 0, 128, 10, 2); break;
                case 5: rc |= WG3(128, 0, 128, 10, 2); break;
#undef WG3
                default: rc |= launch_wgrad(a, ws);
            }
        } else {
            rc |= launch_wgrad(a, ws);
        }
    };
    // ConvTranspose2d(cin, cout, 2, 2): dW[ci][co][d] = sum In[p][ci] * dU[up(p,d)][co]; dU = channels [0,cout) of dcat
    auto wgrad_up_now = [&](int l, const float* in, const float* dcat, int cs_cat, int H) {
        const ConvLayerDesc& d = kConv[l];
        WgradArgs a{};
        a.R = in; a.csR = d.cin0; a.coR = 0;
        a.C0p = dcat; a.C1p = dcat; a.csC0 = cs_cat; a.csC1 = cs_cat; a.nC0 = d.cout;
        a.dW = grads + po.conv_w[l]; a.sM = d.cout * 4; a.sN = 4; a.sT = 1;
        a.kind = UPCONV; a.taps = 4; a.Mb = d.cin0 / 32; a.Nb = d.cout / 32;
        a.nimg = nimg; a.H = H; a.W = H;
        a.partial = G(g.WG);
        a.db = grads + po.conv_b[l]; a.nbias = d.cout;          // bias gradient = column sums of dU over all four taps
        rc |= launch_wgrad(a, ws);
    };
    // Forks are not free (an event record drains the caller's queue between two data gradients, the wait costs the side queue a few
    // microseconds): GIGA_WGRAD_FORK_EVERY = n enqueues the weight gradients in groups of n behind ONE fork (default 1).
    const int fork_every = [] { const char* e = getenv("GIGA_WGRAD_FORK_EVERY"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    std::function<void()> pending[16];
    int npending = 0;
    auto flush = [&]() {
        if (npending == 0) return;
        fork();
        for (int i = 0; i < npending; ++i) pending[i]();
        npending = 0;
    };
    auto defer = [&](std::function<void()> f) {
        pending[npending++] = std::move(f);
        if (npending >= fork_every) flush();
    };
    auto wgrad3 = [&](int l, const float* dpre, const float* in0, const float* in1, int H) {
        defer([=, &wgrad3_now]() { wgrad3_now(l, dpre, in0, in1, H); });
    };
    auto wgrad_up = [&](int l, const float* in, const float* dcat, int cs_cat, int H) {
        defer([=, &wgrad_up_now]() { wgrad_up_now(l, in, dcat, cs_cat, H); });
    };
    // data gradient of layer l; relu_of: the forward activation the gradient flows into next (ReLU output of the layer
    // below): its backward mask is applied in the convolution's epilogue instead of by a separate pass over the tensor
    auto dgrad_args = [&](int l, const float* in, int cs, void* out, const float* relu_of = nullptr) {
        ConvArgs a{};
        a.in0 = in; a.in1 = nullptr; a.w = bwd_blob + (MATH == MATH_BF16 ? bo.convbf[l] : bo.conv[l]); a.bias = nullptr; a.out = out; a.nimg = nimg;
        a.cs0 = cs;
        a.mask = relu_of;
        a.xcd_local = 1;                 // images onto XCDs, all weight groups of an image on one L2 (giga_conv16.h: conv_wg_map)
        return a;
    };
    const size_t n40 = (size_t)nimg * 1600, n20 = (size_t)nimg * 400, n10 = (size_t)nimg * 100;

    // ---- the data-gradient chain.  Stage order L12 L11 L10 L9 L8 L7 L6 L5 L4 [pool1] L3 L2 [pool0] L1 L0: one launch per stage, or
    // ONE persistent launch (unet_dgrad_mega_kernel, giga_encoder.hip).
    BwdMegaArgs M{};
    M.layer[0] = dgrad_args(12, gplanes, 0, G(g.gA6), F(f.A6));          // L12 conv_final (1x1, no activation): gplanes = dOUT
    M.layer[1] = dgrad_args(11, G(g.gA6), 0, G(g.gA5), F(f.A5));         // L11 up1.conv2: A5 -> A6
    M.layer[2] = dgrad_args(10, G(g.gA5), 0, G(g.gC1));                  // L10 up1.conv1: cat(U1, S0) -> A5; 64 channels out (dU1 | dS0 skip part)
    M.layer[3] = dgrad_args(9, G(g.gC1), 64, G(g.gA4), F(f.A4));         // L9 up1.upconv: A4 (20x20x64) -> U1; dU1 = gC1[..., 0:32]
    M.layer[4] = dgrad_args(8, G(g.gA4), 0, G(g.gA3), F(f.A3));          // L8 up0.conv2: A3 -> A4
    M.layer[5] = dgrad_args(7, G(g.gA3), 0, G(g.gC0));                   // L7 up0.conv1: cat(U0, S1) -> A3; 128 channels out
    M.layer[6] = dgrad_args(6, G(g.gC0), 128, G(g.gS2), F(f.S2));        // L6 up0.upconv: S2 (10x10x128) -> U0; dU0 = gC0[..., 0:64]
    M.layer[7] = dgrad_args(5, G(g.gS2), 0, G(g.gA2), F(f.A2));          // L5 down2.conv2: A2 -> S2
    M.layer[8] = dgrad_args(4, G(g.gA2), 0, G(g.gQ1));                   // L4 down2.conv1: Q1 -> A2
    M.pool[0] = BwdPool{G(g.gS1), G(g.gC0), G(g.gQ1), F(f.S1), F(f.Q1), 128, 64, 20, 20, 64};   // dS1 = gC0[..., 64:128] + unpool(dQ1)
    M.layer[9] = dgrad_args(3, G(g.gS1), 0, G(g.gA1), F(f.A1));          // L3 down1.conv2: A1 -> S1
    M.layer[10] = dgrad_args(2, G(g.gA1), 0, G(g.gQ0));                  // L2 down1.conv1: Q0 -> A1
    M.pool[1] = BwdPool{G(g.gS0), G(g.gC1), G(g.gQ0), F(f.S0), F(f.Q0), 64, 32, 40, 40, 32};    // dS0 = gC1[..., 32:64] + unpool(dQ0)
    M.layer[11] = dgrad_args(1, G(g.gS0), 0, G(g.gA0), F(f.A0));         // L1 down0.conv2: A0 -> S0
    M.layer[12] = dgrad_args(0, G(g.gA0), 0, G(g.gP0));                  // L0 down0.conv1: P0 -> A0
    M.sync = reinterpret_cast<unsigned*>(gws + g.total);                  // (behind the workspace proper: enc_bwd_workspace_bytes)
    // The persistent form is an OPT-IN (GIGA_DGRAD_PERSIST=1): measured at 32 scenes it is SLOWER than the fifteen launches it replaces
    // (bf16 step 1.016 -> 1.057 ms, fp32 1.824 -> 1.849: the kernel takes 212 us against 194 us for the separate launches, whose
    // dispatch the queue already hides behind the previous kernel's tail, and fifteen group barriers are not free; the weight gradients,
    // pushed behind the whole chain, find their dY further down the cache hierarchy: +1-2 us each).  profiles/r05/dgrad_mega.txt
    static const int env_dgrad_persist = [] { const char* e = getenv("GIGA_DGRAD_PERSIST"); return e ? atoi(e) : 0; }();
    const int mega = env_dgrad_persist ? launch_unet_dgrad_mega(M, MATH == MATH_BF16, s) : 0;
    if (mega < 0) rc |= mega;
    if (mega > 0) {
        // the weight gradients: every one needs only what the chain has left in memory (dPre of its layer, mask applied) and the
        // forward's activations
        wgrad3(12, gplanes, F(f.A6), nullptr, 40);
        wgrad3(11, G(g.gA6), F(f.A5), nullptr, 40);
        wgrad3(10, G(g.gA5), F(f.U1), F(f.S0), 40);
        wgrad_up(9, F(f.A4), G(g.gC1), 64, 20);
        wgrad3(8, G(g.gA4), F(f.A3), nullptr, 20);
        wgrad3(7, G(g.gA3), F(f.U0), F(f.S1), 20);
        wgrad_up(6, F(f.S2), G(g.gC0), 128, 10);
        wgrad3(5, G(g.gS2), F(f.A2), nullptr, 10);
        wgrad3(4, G(g.gA2), F(f.Q1), nullptr, 10);
        wgrad3(3, G(g.gS1), F(f.A1), nullptr, 20);
        wgrad3(2, G(g.gA1), F(f.Q0), nullptr, 20);
        wgrad3(1, G(g.gS0), F(f.A0), nullptr, 40);
        wgrad3(0, G(g.gA0), F(f.P0), nullptr, 40);
    }
    // stage k of the chain is the data gradient of U-Net layer 12 - k
    static constexpr int LAYER_OF_STAGE[13] = {12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0};
    // fp32 step: the data gradients of the 3x3 layers are stride-1 3x3 convolutions too (flipped taps, channels swapped) and run as
    // Winograd F(2x2, 3x3) on the images giga_derive_winograd leaves behind the backward blob's other regions (giga_wino.h; the ReLU
    // mask of the layer below in the epilogue as before).  GIGA_WINOGRAD_BWD=0: the direct conv16 kernels.
    static const int env_wino_bwd = [] { const char* e = getenv("GIGA_WINOGRAD_BWD"); return e ? atoi(e) : 1; }();
    constexpr bool CAN_WINO_BWD = MATH == MATH_NATIVE;
    const bool wino_bwd = CAN_WINO_BWD && env_wino_bwd != 0;
    auto dgrad3 = [&](auto c0, auto cout, auto hw, auto nb, int stage, int layer) {
        constexpr int C0 = decltype(c0)::value, COUT = decltype(cout)::value, HW = decltype(hw)::value, NB = decltype(nb)::value;
        if constexpr (CAN_WINO_BWD) {
            if (wino_bwd) {
                ConvArgs a = M.layer[stage];
                a.w = bwd_blob + bo.wino[layer];
                return launch_wino<C0, 0, COUT, HW, HW, false, false>(a, s);
            }
        }
        return launch_conv<float, CONV3, C0, 0, COUT, HW, HW, NB, false, false, MATH>(M.layer[stage], s);
    };
    if (mega <= 0) {
        // one launch per stage, the weight gradient of a layer right behind the launch that produced its dPre (it is still in the
        // Infinity Cache then)
        wgrad3(12, gplanes, F(f.A6), nullptr, 40);
        rc |= launch_conv<float, CONV1, 32, 0, 32, 40, 40, 2, false, false, MATH>(M.layer[0], s);
        wgrad3(11, G(g.gA6), F(f.A5), nullptr, 40);
        rc |= dgrad3(IC<32>{}, IC<32>{}, IC<40>{}, IC<2>{}, 1, LAYER_OF_STAGE[1]);
        wgrad3(10, G(g.gA5), F(f.U1), F(f.S0), 40);
        rc |= dgrad3(IC<32>{}, IC<64>{}, IC<40>{}, IC<2>{}, 2, LAYER_OF_STAGE[2]);
        wgrad_up(9, F(f.A4), G(g.gC1), 64, 20);
        rc |= launch_conv<float, DOWN, 32, 0, 64, 20, 20, 2, false, false, MATH>(M.layer[3], s);
        wgrad3(8, G(g.gA4), F(f.A3), nullptr, 20);
        rc |= dgrad3(IC<64>{}, IC<64>{}, IC<20>{}, IC<1>{}, 4, LAYER_OF_STAGE[4]);
        wgrad3(7, G(g.gA3), F(f.U0), F(f.S1), 20);
        rc |= dgrad3(IC<64>{}, IC<128>{}, IC<20>{}, IC<1>{}, 5, LAYER_OF_STAGE[5]);
        wgrad_up(6, F(f.S2), G(g.gC0), 128, 10);
        rc |= launch_conv<float, DOWN, 64, 0, 128, 10, 10, 1, false, false, MATH>(M.layer[6], s);
        wgrad3(5, G(g.gS2), F(f.A2), nullptr, 10);
        rc |= dgrad3(IC<128>{}, IC<128>{}, IC<10>{}, IC<1>{}, 7, LAYER_OF_STAGE[7]);
        wgrad3(4, G(g.gA2), F(f.Q1), nullptr, 10);
        rc |= dgrad3(IC<128>{}, IC<64>{}, IC<10>{}, IC<1>{}, 8, LAYER_OF_STAGE[8]);
        {
            const size_t tot = n20 * 64 / 4;
            GIGA_LAUNCH(pool_bwd_add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, G(g.gS1),
                               G(g.gC0), 128, 64, G(g.gQ1), F(f.S1), F(f.Q1), nimg, 20, 20, 64);
        }
        wgrad3(3, G(g.gS1), F(f.A1), nullptr, 20);
        rc |= dgrad3(IC<64>{}, IC<64>{}, IC<20>{}, IC<1>{}, 9, LAYER_OF_STAGE[9]);
        wgrad3(2, G(g.gA1), F(f.Q0), nullptr, 20);
        rc |= dgrad3(IC<64>{}, IC<32>{}, IC<20>{}, IC<2>{}, 10, LAYER_OF_STAGE[10]);
        {
            const size_t tot = n40 * 32 / 4;
            GIGA_LAUNCH(pool_bwd_add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, G(g.gS0),
                               G(g.gC1), 64, 32, G(g.gQ0), F(f.S0), F(f.Q0), nimg, 40, 40, 32);
        }
        wgrad3(1, G(g.gS0), F(f.A0), nullptr, 40);
        rc |= dgrad3(IC<32>{}, IC<32>{}, IC<40>{}, IC<2>{}, 11, LAYER_OF_STAGE[11]);
        wgrad3(0, G(g.gA0), F(f.P0), nullptr, 40);
        rc |= dgrad3(IC<32>{}, IC<32>{}, IC<40>{}, IC<2>{}, 12, LAYER_OF_STAGE[12]);
    }
    flush();
    rc |= launch_wgrad3_reduce_all(RED, ws);
    // conv_in + projection
    {
        const int nxp = enc_nxp(B);
        const float* cw = reinterpret_cast<const float*>(blob + ko.convin_w);
        const float* cb = reinterpret_cast<const float*>(blob + ko.convin_b);
        float* part = G(g.CINP);
        const uint4* mask = convin_mask ? reinterpret_cast<const uint4*>(fws + f.MASK) : nullptr;
        auto go = [&](auto kern, int sxw, dim3 grid) {
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)cb_lds_bytes(sxw));
            GIGA_LAUNCH(kern, grid, dim3(512), cb_lds_bytes(sxw), s, tsdf, cw, cb, G(g.gP0), part, B, mask);
        };
        if (nxp == 1) { if (mask) go(convin_bwd_kernel<5, true>, 5, dim3(1, 8, B)); else go(convin_bwd_kernel<5, false>, 5, dim3(1, 8, B)); }
        else          { if (mask) go(convin_bwd_kernel<1, true>, 1, dim3(5, 8, B)); else go(convin_bwd_kernel<1, false>, 1, dim3(5, 8, B)); }
        GIGA_LAUNCH(convin_bwd_reduce_kernel, dim3(CB_PART, 2), dim3(256), 0, s, part, B, nxp, grads + po.conv_in_w,
                           grads + po.conv_in_b);
    }
    rc |= side.join();                                // the caller's stream carries every gradient (the side stream's queue is far behind: its work ended with the reduce)
    if (hipGetLastError() != hipSuccess) rc |= -10;
    return rc;
}

int launch_encoder_backward(const float* tsdf, const uint8_t* blob, const uint8_t* bwd_blob, const uint8_t* fws, float* gplanes,
                            uint8_t* gws, float* grads, int head_present, int B, hipStream_t s, bool bf16_convs, bool convin_mask,
                            SideScope& side) {
    return bf16_convs ? encoder_backward_impl<MATH_BF16>(tsdf, blob, bwd_blob, fws, gplanes, gws, grads, head_present, B, s, convin_mask, side)
                      : encoder_backward_impl<MATH_NATIVE>(tsdf, blob, bwd_blob, fws, gplanes, gws, grads, head_present, B, s, convin_mask, side);
}

}  // namespace giga

#ifdef GIGA_TRACE
extern "C" int giga_debug_wgrad3_trace(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_wg3_trace), sizeof(long long) * 8 * 32) == hipSuccess ? 0 : -10;
}
#endif
