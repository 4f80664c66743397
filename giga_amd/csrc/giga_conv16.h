// conv16: the LDS-staged implicit-GEMM convolution used by every U-Net layer (forward) and by the
// data-gradient convolutions of the training path (giga_encoder_bwd.hip).
#pragma once
#include <type_traits>
#include "giga_dev.h"

namespace giga {

// ====================================================================================================
// U-Net convolutions: implicit GEMM on 16x16 MFMA tiles, NHWC activations, wave-independent units.
//   D[pixel][cout] = sum_{tap,cin} X[pixel+tap][cin] * W[cout][cin][tap]
//   A operand = 16 pixels (a 4x4 block = 2x2 quads of 2x2 pixels; D row = 4*quad + pos, so a lane's 4
//   D registers are one complete quad and the fused 2x2 max-pool is in-lane), B operand = packed
//   weight fragments streamed from L2 (identical for every wave).
//   Persistent workgroups of CONV_NW waves, one per CU.  A workgroup owns one WEIGHT GROUP (NB*16 output
//   channels of one sub-output) whose fragments stay resident in LDS for the whole kernel (<= 72 KiB,
//   filled once by LDS-DMA), so B operands are ds_read_b128 at LDS bandwidth instead of 1 KiB per
//   MFMA-quad per wave through the 64 B/clk L1.  Work unit = (image, 4x4 pixel tile); EVERY WAVE IS
//   INDEPENDENT after the weight load: it stages the haloed 6x6 input patch of its tile in a
//   wave-private LDS region (32 channels at a time, 16-byte pad per pixel), runs the MFMA loop and
//   writes its outputs.  No workgroup barrier in the loop, no tile->wave quantisation; units are
//   strided over all waves of the weight group so each SIMD carries the same number of MFMAs.  The
//   next 32-channel chunk is prefetched into registers while the current one is in the MFMA loop.
//   KIND: CONV3 (3x3, pad 1, +bias, ReLU, optional pool), UPCONV (ConvTranspose2d k=2 s=2 as four
//   1x1 GEMMs scattered to (2y+dy, 2x+dx)), CONV1 (1x1, +bias, no activation).
// ====================================================================================================
struct ConvArgs {
    const void* in0; const void* in1;    // NHWC sources (cat order in0 then in1), C0 / C1 channels
    const uint8_t* w;                    // packed fragments for this precision
    const float* bias;
    void* out;                           // NHWC [img][H'][W'][COUT]
    void* out_pool;                      // NHWC [img][H/2][W/2][COUT] (POOL only)
    float* out_nchw;                     // optional fp32 NCHW copy (CONV1 only)
    const float* mask;                   // optional fp32 tensor of the output's shape: out = mask > 0 ? out : 0 (ReLU backward
                                         // of the layer below, fused into a data-gradient convolution; not with UPCONV)
    int nimg;
    int cs0, co0, cs1, co1;              // channel stride / first channel of in0, in1 (0 stride = dense C0 / C1)
    int trace_id;                        // layer index (diagnostic builds)
    int xcd_local;                       // 1: per-layer launches map the images onto the XCDs (conv_wg_map)
};

constexpr int MATH_NATIVE = 0, MATH_SPLIT = 1, MATH_BF16 = 2;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_16_bf(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int CONV_NW = 12;       // waves per workgroup (3 per SIMD; VGPR use is < 80)

#ifdef GIGA_TRACE   // diagnostic build: issue timeline (s_memtime) of workgroup 0 of the layer selected at run time
static __device__ long long g_conv_trace[CONV_NW * 64];
static __device__ int g_conv_trace_layer = -1;
#define CONV_T(idx) do { if (block == 0 && lane == 0 && a.trace_id == g_conv_trace_layer && (idx) < 64) \
        g_conv_trace[wave * 64 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CONV_T(idx) do {} while (0)
#endif

// linear 16-pixel units for the 10x10 layers (not for DOWN, whose H x W is the coarse output grid, and not with POOL)
template <int KIND, int H, int W>
constexpr bool conv_lin() { return H == 10 && W == 10 && KIND != DOWN; }
// waves per workgroup: 12 unless the resident weights + the wave-private patches would not fit the 160 KiB LDS
template <typename T, int KIND, int C0, int C1, int H, int W, int NB, int MAXW = CONV_NW>
constexpr int conv_nw() {       // (SPLIT instantiations have T = float and exactly the fp32 LDS footprint)
    constexpr int ES = (int)sizeof(T), PS = 32 * ES + 16, HALO = KIND == CONV3 ? 1 : 0;
    constexpr int TAPS = KIND == CONV3 ? 9 : KIND == DOWN ? 4 : 1;
    constexpr int NPIX = conv_lin<KIND, H, W>() ? (W + 2 * HALO) * (3 + 2 * HALO) : KIND == DOWN ? 64 : (4 + 2 * HALO) * (4 + 2 * HALO);
    constexpr int REGION = (NPIX * PS + 15) / 16 * 16;
    constexpr int WB = NB * TAPS * ((C0 + C1) / 32 * (ES == 4 ? 2 : 1)) * (int)FRAG;
    constexpr int fit = (160 * 1024 - WB) / REGION;
    return fit >= MAXW ? MAXW : (fit / 2) * 2;   // (MAXW: the persistent kernel of the Winograd path launches 8 waves, giga_wino.h)
}

// H, W are the OUTPUT-grid dimensions for DOWN (its input is 2H x 2W) and the input dimensions otherwise.
// MATH selects the arithmetic for T = float activations (T = _Float16 is always native f16):
//   MATH_NATIVE  fp32 MFMA on fp32 operands (or f16 MFMA when T is _Float16)
//   MATH_SPLIT   f16x3 split-operand arithmetic (below)
//   MATH_BF16    bf16 operands / fp32 accumulate: activations stay fp32 in HBM (so the fp32 weight-gradient and elementwise
//                backward kernels of the training path read them unchanged) and are rounded to bf16 when a patch is staged;
//                weights are bf16 fragments derived on the device from the fp32 ones; one v_mfma_f32_16x16x32_bf16 per tap
//                and 32-channel chunk.  The training step's forward and data-gradient convolutions (BASELINE config c5).
// MATH_SPLIT (T = float only): f16x3 split-operand arithmetic.  Activations stay fp32 in HBM; the staging step converts every
// value v to the pair hi = f16(v), lo = f16(v - hi) (patch pixel = 4 groups of 8 channels x [8 hi | 8 lo] halfs, 128 B as for
// fp32), the resident weights are [hi, lo] fragment pairs, and every (tap, 32-channel chunk) is three v_mfma_f32_16x16x32_f16
// (W_lo*x_hi + W_hi*x_lo + W_hi*x_hi) instead of eight v_mfma_f32_16x16x4_f32: fp32-grade results at 5x less MFMA time.
// (Measured dead end, round 3: shipping the activations between the layers as (hi | lo << 16) words, split once by the producing
// epilogue so that staging only permutes bytes -- 1 VALU instruction per staged value instead of 5 -- changes no layer's time
// (profiles/r03/encoder_stage_times_packed_x3_experiment.txt): the split layers wait on LDS operand reads, 1.3 KiB per 17-clock
// MFMA, not on the conversion.)
// The layer is written as two device functions so that it can run either as its own kernel (conv16_kernel) or as one
// stage of the persistent U-Net kernel (unet_mega_kernel, giga_encoder.hip):
//   conv16_fill : issue the LDS-DMA of this workgroup's weight group (all launched waves take part)
//   conv16_run  : wait for it, then walk the units.  `block` / `nblocks` replace blockIdx.x / gridDim.x; waves beyond the
//                 layer's own wave count (launched because another stage needs them) only take part in the barrier.
// Workgroup -> (weight group, position among the group's workgroups, image range).  Plain map: group = block % NGRP, every group
// walks all images.  XCD-LOCAL map (per-layer launches of large batches): workgroup b runs on XCD b % 8 (the dispatcher's
// round-robin; an offset from the previous launch only renames the XCDs), and each XCD has its own 4-MiB L2.  A unit's haloed
// patch is read by the NGRP workgroups that compute its output-channel groups and, through the halo, by its neighbours: with
// group = b % NGRP those readers sit on different XCDs -- b % 8 fixes b % NGRP, so an XCD computes ONE group for ALL images -- and
// every re-read crosses the fabric (139.5 MB per launch for 29.8 MB of input in up0.conv1, profiles/r02_traffic_c2.json).  Here XCD x
// takes the x-th eighth of the images and its workgroups cover all the groups, so the re-reads hit in that XCD's L2.
struct ConvWgMap { int grp, wg_in_grp, wgs_per_grp, img0, nimg; };
template <int NGRP>
__device__ __forceinline__ ConvWgMap conv_wg_map(const ConvArgs& a, int block, int nblocks) {
    ConvWgMap m;
    if (a.xcd_local && nblocks % (8 * NGRP) == 0 && a.nimg % 8 == 0) {
        const int xcd = block & 7, j = block >> 3;
        m.grp = j % NGRP; m.wg_in_grp = j / NGRP; m.wgs_per_grp = (nblocks >> 3) / NGRP;
        m.nimg = a.nimg >> 3; m.img0 = xcd * m.nimg;
    } else {
        m.grp = block % NGRP; m.wg_in_grp = block / NGRP; m.wgs_per_grp = nblocks / NGRP;
        m.nimg = a.nimg; m.img0 = 0;
    }
    return m;
}

template <typename T, int KIND, int C0, int C1, int COUT, int NB, int MATH>
__device__ __forceinline__ void conv16_fill(const ConvArgs& a, uint8_t* smem, int block, int nblocks) {
    constexpr bool SPLIT = MATH == MATH_SPLIT;
    constexpr int TAPS = KIND == CONV3 ? 9 : KIND == DOWN ? 4 : 1;
    constexpr int KGT = (C0 + C1) / 32 * ((sizeof(T) == 2 || MATH != MATH_NATIVE) ? 1 : 2);
    constexpr int WPF = SPLIT ? 2 : 1;
    constexpr int NSUB = KIND == UPCONV ? 4 : 1, NBT = COUT / 16, CG = NBT / NB, NGRP = NSUB * CG;
    constexpr int WFRAGS = NB * TAPS * KGT * WPF;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwl = blockDim.x >> 6;
    const int grp = conv_wg_map<NGRP>(a, block, nblocks).grp;
    const int sub = grp / CG, nb0 = (grp % CG) * NB;
    // fragments of (sub, nb0 .. nb0+NB-1) are contiguous in the packed blob: [sub][nb][tap][kg]
    const uint8_t* wsrc = a.w + (size_t)(sub * NBT + nb0) * TAPS * KGT * WPF * FRAG;
    for (int c = wave; c < WFRAGS; c += nwl)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wsrc + (size_t)c * FRAG + lane * 16),
            (__attribute__((address_space(3))) void*)(smem + c * FRAG), 16, 0, 0);
}

template <typename T, int KIND, int C0, int C1, int COUT, int H, int W, int NB, bool POOL, bool RELU, int MATH, int MAXW = CONV_NW>
__device__ __forceinline__ void conv16_run(const ConvArgs& a, uint8_t* smem, int block, int nblocks) {
    constexpr bool SPLIT = MATH == MATH_SPLIT, BF = MATH == MATH_BF16;
    static_assert(MATH == MATH_NATIVE || sizeof(T) == 4, "split / bf16 modes read and write fp32 activations");
    constexpr int CIN = C0 + C1;
    constexpr int TAPS = KIND == CONV3 ? 9 : KIND == DOWN ? 4 : 1;
    constexpr int HALO = KIND == CONV3 ? 1 : 0;
    constexpr int NCHUNK = CIN / 32;
    constexpr int ES = (int)sizeof(T);
    constexpr int PS = BF ? 32 * 2 + 16 : 32 * ES + 16;   // LDS pixel stride in bytes (bf16 patches: 64 B of channels)
    // LIN (the 10x10 layers): a 4x4 tiling covers 144 pixel slots for 100 pixels; instead a unit is 16 CONSECUTIVE
    // pixels of an image in row-major order (7 units per image), staged as the haloed band of rows they touch.
    constexpr bool LIN = conv_lin<KIND, H, W>();
    constexpr int NWV = conv_nw<T, KIND, C0, C1, H, W, NB, MAXW>();
    constexpr int LW = LIN ? W + 2 * HALO : KIND == DOWN ? 8 : 4 + 2 * HALO;                 // staged patch width
    constexpr int LH = LIN ? 3 + 2 * HALO : LW;                                                // 16 pixels of a 10-wide image touch <= 3 rows
    constexpr int NPIX = LW * LH;                     // (DOWN: the 8x8 fine pixels)
    constexpr int VPP = 32 * ES / 16;                 // 16-byte vectors per pixel per chunk
    constexpr int NVEC = NPIX * VPP;                  // vectors per chunk
    constexpr int NLD = (NVEC + 63) / 64;             // staging loads per lane
    constexpr int REGION = (NPIX * PS + 15) / 16 * 16;
    constexpr int TX = LIN ? (H * W + 15) / 16 : (W + 3) / 4, TY = LIN ? 1 : (H + 3) / 4;   // LIN: TX = units per image
    constexpr int NSUB = KIND == UPCONV ? 4 : 1;
    constexpr int NBT = COUT / 16;                    // 16-channel blocks per sub-output
    constexpr int CG = NBT / NB;                      // channel groups per sub-output
    constexpr int NGRP = NSUB * CG;                   // weight groups (one per workgroup)
    constexpr bool F16MATH = ES == 2 || SPLIT || BF;  // 16x16x32 f16 / bf16 MFMA (one k-group per 32-channel chunk)
    constexpr int KGC = F16MATH ? 1 : 2;              // k-groups per 32-channel chunk (16 / 32 channels)
    constexpr int WPF = SPLIT ? 2 : 1;                // LDS fragments per weight k-group ([hi, lo] pair)
    constexpr int KGT = CIN / 32 * KGC;
    constexpr int WFRAGS = NB * TAPS * KGT * WPF;     // weight fragments resident in LDS
    static_assert(NBT % NB == 0, "cout grouping");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const bool active = wave < NWV;                    // (the persistent kernel launches 12 waves for every stage)
    uint8_t* region = smem + (size_t)WFRAGS * FRAG + (active ? wave : 0) * REGION;
    CONV_T(0);

    // this workgroup's weight group (filled by conv16_fill)
    const ConvWgMap wm = conv_wg_map<NGRP>(a, block, nblocks);
    const int grp = wm.grp, wg_in_grp = wm.wg_in_grp, wgs_per_grp = wm.wgs_per_grp;
    const int sub = grp / CG, nb0 = (grp % CG) * NB;
    const uint4* wl = reinterpret_cast<const uint4*>(smem);
    int tcount = 2;

    const int nwaves = wgs_per_grp * NWV;
    const int units = wm.nimg * TY * TX;               // units of this workgroup's image range

    // A geometry: row i = lane&15 : quad = i>>2 (qy = quad>>1, qx = quad&1), pos = i&3 (dy = pos>>1, dx = pos&1)
    const int ay = 2 * (j >> 3) + ((j >> 1) & 1), ax = 2 * ((j >> 2) & 1) + (j & 1);
    constexpr int AG = SPLIT ? 32 : 16;               // byte offset of k-group g inside a patch pixel
    int a_off = (KIND == DOWN ? (2 * ay * LW + 2 * ax) : (ay * LW + ax)) * PS + g * AG;      // LIN: set per unit below

    // staging geometry of this lane's NLD vectors (fixed for the whole kernel)
    int st_lds[NLD], st_ly[NLD], st_lx[NLD], st_v[NLD], st_pl[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int i = lane + 64 * q;
        const int pix = i / VPP;
        st_v[q] = i % VPP;
        st_ly[q] = pix / LW; st_lx[q] = pix % LW;
        st_lds[q] = i < NVEC ? pix * PS + st_v[q] * 16 : -1;
        st_pl[q] = st_ly[q] * (KIND == DOWN ? 2 * W : W) + st_lx[q];       // pixel offset from the patch origin in the input image
    }
    auto unit_coords = [&](int u, int& tx, int& ty, int& img) {
        tx = u % TX; ty = (u / TX) % TY; img = wm.img0 + u / (TX * TY);
    };
    // Patch loads.  fp32-input MFMAs do not co-execute with VALU work of any wave of the SIMD (DESIGN "fp32 MFMA and the
    // VALU"), so every VALU instruction of the loop is paid in MFMA time: the bounds tests and the pixel -> byte-offset
    // arithmetic are done ONCE PER UNIT (unit_setup: 24-bit pixel indices, 32-bit byte offsets from the tensor base; launch_conv
    // refuses tensors beyond that, GIGA_MAX_SCENES keeps callers inside), a chunk's loads are `uniform base + per-lane offset` with no arithmetic at all, out-of-image lanes load
    // from offset 0 and are simply not written to the patch, whose out-of-image cells are zeroed once per unit.
    uint4 stg[NLD];
    uint32_t st_b0[NLD], st_b1[C1 > 0 ? NLD : 1];      // byte offsets of this lane's vectors from in0 / in1 (0 if outside)
    bool st_ok[NLD];                                   // inside the image (and inside the patch)
    const uint32_t rowb0 = (uint32_t)(a.cs0 ? a.cs0 : C0) * ES, rowb1 = (uint32_t)(a.cs1 ? a.cs1 : C1) * ES;
    auto unit_setup = [&](int u) {
        int tx, ty, img;
        unit_coords(u, tx, ty, img);
        constexpr int IH = KIND == DOWN ? 2 * H : H, IW = KIND == DOWN ? 2 * W : W;   // input grid
        constexpr int TS = KIND == DOWN ? 8 : 4;                                        // input pixels per tile edge
        const int y0 = (LIN ? (16 * tx) / W : TS * ty) - HALO, x0 = (LIN ? 0 : TS * tx) - HALO;
        const int pix0 = (img * IH + y0) * IW + x0;                                     // (uniform)
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int gy = y0 + st_ly[q], gx = x0 + st_lx[q];
            const bool ok = st_lds[q] >= 0 && (unsigned)gy < (unsigned)IH && (unsigned)gx < (unsigned)IW;
            const uint32_t pix = (uint32_t)(pix0 + st_pl[q]);            // < 2^24 (launch_conv)
            st_ok[q] = ok;
            st_b0[q] = ok ? __umul24(pix, rowb0) + st_v[q] * 16 : 0u;
            if constexpr (C1 > 0) st_b1[q] = ok ? __umul24(pix, rowb1) + st_v[q] * 16 : 0u;
        }
    };
    auto issue_loads = [&](int cc) {
        const bool first = C1 == 0 || cc * 32 < C0;
        const char* base = first ? reinterpret_cast<const char*>(a.in0) + (size_t)(a.co0 + cc * 32) * ES
                                 : reinterpret_cast<const char*>(a.in1) + (size_t)(a.co1 + cc * 32 - C0) * ES;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            uint32_t off = st_b0[q];
            if constexpr (C1 > 0) off = first ? st_b0[q] : st_b1[q];
            stg[q] = *reinterpret_cast<const uint4*>(base + off);
        }
    };

    // global wave index = wave * (#workgroups) + workgroup: remainder units spread over all CUs/SIMDs.
    // TAIL SPLIT (fp32 3x3 layers): with U units over nwaves waves the last round holds only U mod nwaves units, i.e. one
    // lone wave on one or two SIMDs of every CU walks a whole unit -- NCHUNK dependent load -> stage -> MFMA rounds, each a
    // global-load latency long because nobody shares its SIMD -- while the other SIMDs idle (at 32 scenes, 3.125 rounds: up to
    // 15 % of the layer).  When the remainder is at most NWV/TPARTS units per workgroup, each of them is cut into TPARTS parts that
    // run side by side on TPARTS waves (wave = TPARTS*slot + part, i.e. different SIMDs): one part per 32-channel chunk, or, for
    // the 32-channel layers, one part per kernel row.  Parts > 0 hand their partial sums over through their own (dead) patch
    // regions; part 0 adds them in order, ((p0 + p1) + p2) + p3, and runs the epilogue.  One workgroup barrier, taken by
    // every wave of the workgroup.
    constexpr int TPARTS = NCHUNK == 1 ? 3 : NCHUNK;  // parts of a split unit
    constexpr bool TSPLIT_OK = KIND == CONV3 && MATH == MATH_NATIVE && ES == 4 && NB * 1024 <= REGION && NWV >= TPARTS && TPARTS <= 4;
    const int full_end = (units / nwaves) * nwaves, rem = units - full_end;
    const int smax = (rem + wgs_per_grp - 1) / wgs_per_grp;
    const bool tsplit = TSPLIT_OK && rem > 0 && TPARTS * smax <= NWV;
    const int lim = tsplit ? full_end : units;       // units below `lim` are walked as whole units
    int tail_u = -1, part = 0;
    if (tsplit) {
        const int slot = wave / TPARTS, ru = slot * wgs_per_grp + wg_in_grp;
        part = wave - TPARTS * slot;
        if (active && ru < rem) tail_u = full_end + ru;
    }
    const int tail_cc = NCHUNK == 1 ? 0 : part;      // the chunk a part works on
    int u = wave * wgs_per_grp + wg_in_grp;
    bool tail = false;
    int cc = 0;
    if (u >= lim) { u = tail_u; tail = true; cc = tail_cc; }
    const bool work = active && u >= 0;
    // the first patch and the biases are requested while the weight fill is still in flight
    if (work) { unit_setup(u); issue_loads(cc); }
    bool fresh = true;                                 // the staged vectors belong to a unit whose patch is not zero-padded yet
    float bias_r[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bias_r[n] = a.bias ? a.bias[(nb0 + n) * 16 + j] : 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): weights (LDS-DMA), patch and biases landed
    __syncthreads();
    CONV_T(1);
    if (!work) {                                       // (no workgroup barrier below this point, except the tail split's)
        if (tsplit) __syncthreads();
        return;
    }
    constexpr int NACC = NB == 1 ? 2 : 1;          // independent accumulator chains per channel block
    f32x4v acc[NB][NACC];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[n][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // Epilogue addressing is 32-bit (byte offsets from the uniform tensor bases; launch_conv bounds the tensors) and uses
    // 24-bit multiplies: like the patch loads, its VALU instructions are paid in MFMA time.
    auto epilogue = [&](const int u) {
        int tx, ty, img;
        unit_coords(u, tx, ty, img);
        char* out = reinterpret_cast<char*>(a.out);
        const int qy = 4 * ty + 2 * (g >> 1), qx = 4 * tx + 2 * (g & 1);
        const int lp0 = 16 * tx + 4 * g;               // LIN: this lane's registers are pixels lp0 .. lp0+3 of the image
        const bool qok = LIN ? lp0 < H * W : (qy < H && qx < W);   // (H, W even / H*W % 4 == 0: all four in or out)
        // pixel index (in the output tensor) of register e
        uint32_t pe[4];
        constexpr int OW = KIND == UPCONV ? 2 * W : W, OHW = KIND == UPCONV ? 4 * H * W : H * W;
        const uint32_t img_pix = (uint32_t)img * OHW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (LIN) {
                if constexpr (KIND == UPCONV) {
                    const int y = (lp0 + e) / W, x = (lp0 + e) % W;
                    pe[e] = img_pix + __umul24(2 * y + (sub >> 1), OW) + 2 * x + (sub & 1);
                } else {
                    pe[e] = img_pix + lp0 + e;
                }
            } else if constexpr (KIND == UPCONV) {
                pe[e] = img_pix + __umul24(2 * qy + (sub >> 1), OW) + 2 * qx + (sub & 1) + (e >> 1) * 2 * OW + (e & 1) * 2;
            } else {
                pe[e] = img_pix + __umul24(qy, OW) + qx + (e >> 1) * OW + (e & 1);
            }
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int co = (nb0 + n) * 16 + j;
            const float bv = bias_r[n];
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sacc = acc[n][0][e];
                if (NACC == 2) sacc += acc[n][NACC - 1][e];
                v[e] = sacc + bv;
                if (RELU) v[e] = relu(v[e]);
                acc[n][0][e] = 0.f;
                acc[n][NACC - 1][e] = 0.f;
            }
            if (qok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t idx = pe[e] * COUT + co;
                    if constexpr (KIND != UPCONV && !RELU) {      // (the data-gradient convolutions have no activation of their own)
                        if (a.mask && !(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.mask) + idx * 4u) > 0.f))
                            v[e] = 0.f;
                    }
                    store_saddr(reinterpret_cast<T*>(out), idx * (uint32_t)ES, (T)v[e]);
                    if constexpr (KIND == CONV1) {
                        if (a.out_nchw) {
                            const uint32_t pin = pe[e] - img_pix;                      // y * W + x
                            a.out_nchw[((size_t)img * COUT + co) * H * W + pin] = v[e];
                        }
                    }
                }
                if (POOL) {
                    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    const uint32_t pidx = ((uint32_t)img * (H / 2 * (W / 2)) + __umul24(qy >> 1, W / 2) + (qx >> 1)) * COUT + co;
                    store_saddr(reinterpret_cast<T*>(a.out_pool), pidx * (uint32_t)ES, (T)mx);
                }
            }
        }
    };

    while (true) {
        if constexpr (LIN) {           // row i of the A operand is pixel 16*tile + i of the image (clamped past the end)
            const int t16 = 16 * (u % TX);
            int p = t16 + j;
            p = p < H * W ? p : H * W - 1;
            a_off = ((p / W - t16 / W) * LW + p % W) * PS + g * AG;
        }
        // ---- registers -> wave-private LDS patch (DS ops of one wave execute in order) -----------
        auto put = [&](int q, const uint4 val) {
            if constexpr (BF) {                       // channels 4v..4v+3 -> four bf16 (round to nearest even), 8 bytes
                const f32x4v x = __builtin_bit_cast(f32x4v, val);
                const bf16x4 b4 = {(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
                *reinterpret_cast<bf16x4*>(region + (st_lds[q] - st_v[q] * 16) + st_v[q] * 8) = b4;
            } else if constexpr (SPLIT) {
                // vector v = channels 4v..4v+3 of the pixel: hi halfs at group (v>>1), slot 4*(v&1); lo 16 bytes further
                const f32x4v x = __builtin_bit_cast(f32x4v, val);
                half4 hi4, lo4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const half_t h = (half_t)x[e];
                    hi4[e] = h;
                    lo4[e] = (half_t)__builtin_fmaf((float)h, -1.0f, x[e]);
                }
                uint8_t* dst = region + (st_lds[q] - st_v[q] * 16) + (st_v[q] >> 1) * 32 + (st_v[q] & 1) * 8;
                *reinterpret_cast<half4*>(dst) = hi4;
                *reinterpret_cast<half4*>(dst + 16) = lo4;
            } else {
                *reinterpret_cast<uint4*>(region + st_lds[q]) = val;
            }
        };
        if (fresh) {                                   // first chunk of a unit: zero its out-of-image cells (border tiles only)
#pragma unroll
            for (int q = 0; q < NLD; ++q)
                if (st_lds[q] >= 0 && !st_ok[q]) put(q, make_uint4(0, 0, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q)
            if (st_ok[q]) put(q, stg[q]);
        CONV_T(tcount); ++tcount;          // patch chunk in LDS (includes the wait for its global loads)
        // ---- prefetch the next chunk / next unit ---------------------------------------------------
        int un = u, ccn = cc + 1;
        bool tailn = tail, more = !tail;             // (a part is one chunk and always a wave's last item)
        if (more && ccn == NCHUNK) {
            ccn = 0;
            un = u + nwaves;
            if (un >= lim) { un = tail_u; tailn = true; ccn = tail_cc; more = un >= 0; }
        }
        fresh = more && un != u;
        if (more) {
            if (fresh) unit_setup(un);
            issue_loads(ccn);
        }
        // ---- MFMA over taps x k-groups of this chunk: A from the patch, B from the resident weights.  The
        // operands of step i+3 are read from LDS right after the MFMAs of step i are issued (software pipeline), so a
        // wave keeps the MFMA pipe fed on its own instead of relying on its two SIMD siblings to cover the
        // LDS round trip (they are gone in the ragged last round).
        auto mfma_block = [&](auto nt_tag, const int a_base, const int wtap0) {
        constexpr int NT = decltype(nt_tag)::value, NIT = NT * KGC;
        auto read_ops = [&](int it, uint4 (&av)[WPF], uint4 (&bw)[NB][WPF]) {
            const int tap = it / KGC, kg = it % KGC;
            const int toff = (KIND == DOWN ? ((tap >> 1) * LW + (tap & 1)) : ((tap / 3) * LW + (tap % 3))) * PS;   // (tap < NT)
#pragma unroll
            for (int w = 0; w < WPF; ++w) av[w] = *reinterpret_cast<const uint4*>(region + a_base + toff + kg * 64 + w * 16);
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int w = 0; w < WPF; ++w) bw[n][w] = wl[(((n * TAPS + wtap0 + tap) * KGT + cc * KGC + kg) * WPF + w) * 64 + lane];
        };
        // ring of PD operand sets: the LDS round trip of a 1 KiB ds_read_b128 is ~300 cycles, i.e. more than two
        // groups of four 32-cycle MFMAs
        constexpr int PD = NIT < 3 ? NIT : 3;
        uint4 av_q[PD][WPF], bw_q[PD][NB][WPF];
#pragma unroll
        for (int it = 0; it < PD; ++it) read_ops(it, av_q[it], bw_q[it]);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int kg = it % KGC, sl = it % PD;
            if constexpr (SPLIT) {
                const half8 Ah = __builtin_bit_cast(half8, av_q[sl][0]), Al = __builtin_bit_cast(half8, av_q[sl][1]);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const half8 Bh = __builtin_bit_cast(half8, bw_q[sl][n][0]), Bl = __builtin_bit_cast(half8, bw_q[sl][n][1]);
                    f32x4v& c = acc[n][it & (NACC - 1)];
                    c = mfma16_16(Ah, Bl, c);
                    c = mfma16_16(Al, Bh, c);
                    c = mfma16_16(Ah, Bh, c);
                }
            } else if constexpr (BF) {
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[n][it & (NACC - 1)] = mfma16_16_bf(__builtin_bit_cast(bf16x8, av_q[sl][0]),
                                                           __builtin_bit_cast(bf16x8, bw_q[sl][n][0]), acc[n][it & (NACC - 1)]);
            } else if constexpr (ES == 2) {
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[n][kg & (NACC - 1)] = mfma16_16(__builtin_bit_cast(half8, av_q[sl][0]),
                                                        __builtin_bit_cast(half8, bw_q[sl][n][0]), acc[n][kg & (NACC - 1)]);
            } else {
                // 16x16x4 f32: 32-cycle issue, 40-cycle dependent latency -> alternate accumulators
                const f32x4v A = __builtin_bit_cast(f32x4v, av_q[sl][0]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[n][e & (NACC - 1)] = mfma32_16(A[e], __builtin_bit_cast(f32x4v, bw_q[sl][n][0])[e], acc[n][e & (NACC - 1)]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (it + PD < NIT) read_ops(it + PD, av_q[sl], bw_q[sl]);
            __builtin_amdgcn_sched_barrier(0);
        }
        };
        if constexpr (TSPLIT_OK && NCHUNK == 1) {
            if (tail) mfma_block(std::integral_constant<int, 3>{}, a_off + part * LW * PS, 3 * part);
            else mfma_block(std::integral_constant<int, TAPS>{}, a_off, 0);
        } else {
            mfma_block(std::integral_constant<int, TAPS>{}, a_off, 0);
        }
        CONV_T(tcount); ++tcount;          // MFMAs of this chunk issued
        // ---- epilogue after the last chunk: lane holds cout j of quad g (4 pixels) ------------------
        if (cc == NCHUNK - 1 && !tail) epilogue(u);
        if (!more) break;
        u = un; cc = ccn; tail = tailn;
    }
    if constexpr (TSPLIT_OK) {
        if (tsplit) {
            float4* mine = reinterpret_cast<float4*>(region);
            if (tail && part != 0) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x4v sacc = acc[n][0];
                    if (NACC == 2) sacc += acc[n][NACC - 1];
                    mine[n * 64 + lane] = make_float4(sacc[0], sacc[1], sacc[2], sacc[3]);
                }
            }
            __syncthreads();
            if (tail && part == 0) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x4v sacc = acc[n][0];
                    if (NACC == 2) sacc += acc[n][NACC - 1];
#pragma unroll
                    for (int t = 1; t < TPARTS; ++t) {
                        const float4 o = reinterpret_cast<const float4*>(region + t * REGION)[n * 64 + lane];
                        sacc += f32x4v{o.x, o.y, o.z, o.w};
                    }
                    acc[n][0] = sacc;
                    if (NACC == 2) acc[n][NACC - 1] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                epilogue(u);
            }
        }
    }
    CONV_T(63);
}

template <typename T, int KIND, int C0, int C1, int COUT, int H, int W, int NB, bool POOL, bool RELU = (KIND == CONV3),
          int MATH = MATH_NATIVE>
__global__ __launch_bounds__((conv_nw<T, KIND, C0, C1, H, W, NB>() * 64)) void conv16_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    conv16_fill<T, KIND, C0, C1, COUT, NB, MATH>(a, smem, (int)blockIdx.x, (int)gridDim.x);
    conv16_run<T, KIND, C0, C1, COUT, H, W, NB, POOL, RELU, MATH>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// LDS bytes of one layer (weights resident + wave-private patches)
template <typename T, int KIND, int C0, int C1, int COUT, int H, int W, int NB, int MATH, int MAXW = CONV_NW>
constexpr size_t conv_lds_bytes() {       // (sized for the fp32 / split footprint; the bf16 mode needs less)
    constexpr bool SPLIT = MATH == MATH_SPLIT;
    constexpr int HALO = KIND == CONV3 ? 1 : 0;
    constexpr int TAPS = KIND == CONV3 ? 9 : KIND == DOWN ? 4 : 1;
    constexpr int ES = (int)sizeof(T);
    constexpr int PS = 32 * ES + 16;
    constexpr bool LIN = conv_lin<KIND, H, W>();
    constexpr int NWV = conv_nw<T, KIND, C0, C1, H, W, NB, MAXW>();
    constexpr int NPIX = LIN ? (W + 2 * HALO) * (3 + 2 * HALO) : KIND == DOWN ? 64 : (4 + 2 * HALO) * (4 + 2 * HALO);
    constexpr int REGION = (NPIX * PS + 15) / 16 * 16;
    constexpr int KGT = (C0 + C1) / 32 * ((ES == 4 && !SPLIT) ? 2 : 1);
    return (size_t)NB * TAPS * KGT * (SPLIT ? 2 : 1) * FRAG + NWV * REGION;
}

template <typename T, int KIND, int C0, int C1, int COUT, int H, int W, int NB, bool POOL, bool RELU = (KIND == CONV3),
          int MATH = MATH_NATIVE>
inline int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr bool SPLIT = MATH == MATH_SPLIT;
    constexpr int HALO = KIND == CONV3 ? 1 : 0;
    constexpr int TAPS = KIND == CONV3 ? 9 : KIND == DOWN ? 4 : 1;
    constexpr int ES = (int)sizeof(T);
    constexpr int PS = 32 * ES + 16;
    constexpr bool LIN = conv_lin<KIND, H, W>();
    constexpr int NWV = conv_nw<T, KIND, C0, C1, H, W, NB>();
    constexpr int NPIX = LIN ? (W + 2 * HALO) * (3 + 2 * HALO) : KIND == DOWN ? 64 : (4 + 2 * HALO) * (4 + 2 * HALO);
    constexpr int REGION = (NPIX * PS + 15) / 16 * 16;
    constexpr int NSUB = KIND == UPCONV ? 4 : 1;
    constexpr int NGRP = NSUB * (COUT / 16 / NB);
    constexpr int KGT = (C0 + C1) / 32 * ((ES == 4 && !SPLIT) ? 2 : 1);
    constexpr size_t lds = (size_t)NB * TAPS * KGT * (SPLIT ? 2 : 1) * FRAG + NWV * REGION;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(!(LIN && POOL), "the in-lane 2x2 max-pool needs the quad tiling");
    static_assert(256 % NGRP == 0, "weight groups must divide the CU count");
    // patch loads address their sources with 32-bit byte offsets (conv16_run::unit_setup)
    // and 24-bit pixel indices; the epilogue stores likewise
    const size_t in_pix = (size_t)a.nimg * (KIND == DOWN ? 4 : 1) * H * W, out_pix = (size_t)a.nimg * (KIND == UPCONV ? 4 : 1) * H * W;
    if (in_pix >= (1u << 24) || in_pix * (size_t)((a.cs0 ? a.cs0 : C0) * ES) >= (1ull << 32) ||
        (C1 > 0 && in_pix * (size_t)((a.cs1 ? a.cs1 : C1) * ES) >= (1ull << 32)) ||
        out_pix * COUT * 4 >= (1ull << 32)) return -7;
    const int units = a.nimg * (LIN ? (H * W + 15) / 16 : ((H + 3) / 4) * ((W + 3) / 4));   // per weight group
    int wgs = (units + NWV - 1) / NWV;                                // workgroups per weight group
    if (wgs > 256 / NGRP) wgs = 256 / NGRP;                           // one persistent workgroup per CU
    else if (a.xcd_local && a.nimg % 8 == 0 && wgs % 8 != 0 && wgs + 8 - wgs % 8 <= 256 / NGRP) wgs += 8 - wgs % 8;   // (a multiple of 8 per group: conv_wg_map)
    auto kern = conv16_kernel<T, KIND, C0, C1, COUT, H, W, NB, POOL, RELU, MATH>;
    if (lds > 48 * 1024)
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
    GIGA_LAUNCH(kern, dim3(wgs * NGRP), dim3(NWV * 64), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
