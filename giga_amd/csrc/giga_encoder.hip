// Encoder kernels: TSDF (B,40,40,40) -> three 40x40x32 feature planes (NHWC).
//
// Replaces reference LocalVoxelEncoder.forward (ConvONets/encoder/voxels.py:89-121):
//   relu(Conv3d(1,32,3,pad=1)) -> permute -> 3x [normalize_coordinate, coordinate2index,
//   torch_scatter.scatter_mean] -> shared UNet (encoder/unet.py:225-239).
// Kernel 1 (convin_project_kernel) fuses the 3-D conv, the ReLU and the three axis means, so the
// 32x40^3 feature volume (8.2 MB/scene in the reference) is never written to memory.  Kernel 2
// (conv_mfma_kernel) is one LDS-tiled implicit-GEMM convolution used for every U-Net layer.
#include "giga_dev.h"

namespace giga {

// ====================================================================================================
// conv_in + ReLU + axis means.
//   grid (NSLAB, B, 2), block 320 = 5 waves.  The workgroup walks SX = 40/NSLAB consecutive ix slices
//   of one iy-half (20 rows) of a scene.  For one slice, wave w owns a strip of 4 iy rows x 40 iz =
//   5 MFMA tiles of (4 iy x 8 iz) voxels.  GEMM per tile: D[voxel][channel] = A[voxel][tap] * Wt[tap][channel],
//   K = 27 taps (+1 zero) = 14 x v_mfma_f32_32x32x2_f32 (exact fp32).
//   The D-row -> voxel map is chosen so that every reduction the projection needs is in-lane:
//     row = (r&3) + 8*(r>>2) + 4*hi ;  iz_local = r & 7 ,  iy_local = 2*(r>>3) + hi
//   * mean over iz (plane 'xy')  : in-lane sum of 8 regs x 5 tiles                -> written directly
//   * mean over iy (plane 'xz')  : in-lane + one cross-half add, fixed-order 5-wave LDS sum, then one
//                                  partial per iy-half to HBM
//   * mean over ix (plane 'yz')  : register accumulation over the slab, one partial per slab to HBM
//   plane_finalize_kernel sums the 2 (xz) / NSLAB (yz) partials in fixed order: deterministic, no atomics.
// Plane pixel (H,W) conventions (common.py:246-251,303-318): xz -> [iz][ix], xy -> [iy][ix], yz -> [iz][iy].
// ====================================================================================================
constexpr int CI_ROWSTRIDE = 56;                  // floats per LDS row: 56 mod 32 = 24 -> the 4 iy rows of a tile hit disjoint banks
constexpr int CI_ROWS = 22;                       // 20 iy rows + halo
constexpr int CI_SLICE = CI_ROWS * CI_ROWSTRIDE;  // one haloed half-slice
constexpr int CI_LDS_SLICES = 3 * CI_SLICE;       // ring of 3 slices (floats)
constexpr int CI_LDS_RED = 5 * 40 * 32;           // cross-wave reduction buffer (floats)
constexpr size_t CI_LDS_BYTES = (CI_LDS_SLICES + CI_LDS_RED) * sizeof(float);

template <typename TOut>
__global__ __launch_bounds__(320) void convin_project_kernel(
    const float* __restrict__ tsdf,        // [B][40][40][40]
    const float* __restrict__ wpk,         // [14][64] packed B operands
    const float* __restrict__ bias,        // [32]
    TOut* __restrict__ planes,             // [3][B][40][40][32] NHWC (xy written here)
    float* __restrict__ xz_partial,        // [2][B][40(iz)][40(ix)][32]   sums over 20 iy
    float* __restrict__ yz_partial,        // [NSLAB][B][40(iz)][40(iy)][32] sums over SX ix
    int B, int SX) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* slices = lds;
    float* red = lds + CI_LDS_SLICES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int slab = blockIdx.x, b = blockIdx.y, half = blockIdx.z;
    const int ix0 = slab * SX, iy0 = half * 20;
    const float* vol = tsdf + (size_t)b * RES * RES * RES;

    for (int i = tid; i < CI_LDS_SLICES; i += blockDim.x) slices[i] = 0.f;   // halos stay zero
    float wreg[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) wreg[s] = wpk[s * 64 + lane];
    const float bn = bias[n];
    __syncthreads();

    // ring slot (ix+1) % 3 holds rows iy0-1 .. iy0+20 of slice ix (zeros outside the volume)
    auto load_slice = [&](int ix) {
        float* dst = slices + ((ix + 1) % 3) * CI_SLICE;
        const bool in = ix >= 0 && ix < RES;
        for (int i = tid; i < CI_ROWS * RES; i += blockDim.x) {
            const int ly = i / RES, z = i % RES;
            const int y = iy0 - 1 + ly;
            dst[ly * CI_ROWSTRIDE + (z + 1)] =
                (in && y >= 0 && y < RES) ? vol[((size_t)ix * RES + y) * RES + z] : 0.f;
        }
    };
    load_slice(ix0 - 1);
    load_slice(ix0);

    // A-operand geometry: M-row i = lane&31 -> (iy_l, iz_l) with i bits (b0,b1,b3)->iz, (b2,b4)->iy
    const int iz_l = (n & 3) | (((n >> 3) & 1) << 2);
    const int iy_l = ((n >> 2) & 1) | (((n >> 4) & 1) << 1);
    const int a_base = (wave * 4 + iy_l) * CI_ROWSTRIDE + iz_l;   // + tile*8 + tap offset (halo origin)

    f32x16 acc_yz[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_yz[t][r] = 0.f;

    const float inv = 1.0f / RES;
    const size_t img_stride = (size_t)RES * RES * CD;
    TOut* plane_xy = planes + ((size_t)1 * B + b) * img_stride;
    float* part_xz = xz_partial + ((size_t)half * B + b) * img_stride;

    for (int sx = 0; sx < SX; ++sx) {
        const int ix = ix0 + sx;
        __syncthreads();                  // everyone finished reading the slot about to be overwritten
        load_slice(ix + 1);
        __syncthreads();
        // slice ix-1+dx lives in ring slot (ix+dx) % 3 (wave-uniform -> SGPRs)
        const int oslot[3] = {((ix + 0) % 3) * CI_SLICE, ((ix + 1) % 3) * CI_SLICE, ((ix + 2) % 3) * CI_SLICE};
        float sum_z[2] = {0.f, 0.f};      // sum over iz for iy_local = 2*g + hi, g = 0,1
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                // this lane supplies tap 2s+hi of K-step s; tap = dx*9 + dy*3 + dz (tap 27: zero weight)
                const int ta = 2 * s, tb = 2 * s + 1 > 26 ? 26 : 2 * s + 1;
                const int offa = oslot[ta / 9] + ((ta / 3) % 3) * CI_ROWSTRIDE + ta % 3 + t * 8;
                const int offb = oslot[tb / 9] + ((tb / 3) % 3) * CI_ROWSTRIDE + tb % 3 + t * 8;
                const float av = slices[a_base + (hi ? offb : offa)];
                d = mfma32(av, wreg[s], d);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = relu(d[r] + bn);
                acc_yz[t][r] += v;
                sum_z[r >> 3] += v;
                d[r] = v;
            }
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                float v = d[z] + d[8 + z];                       // iy_local 0+hi and 2+hi
                v += __shfl_xor(v, 32);                          // other half-wave: the remaining two rows
                if (hi == 0) red[(wave * 40 + t * 8 + z) * 32 + n] = v;      // sum over this wave's 4 iy rows
            }
            // one tile at a time: the 14-MFMA chain is issue-bound (64 cyc each), cross-tile
            // interleaving buys nothing and only multiplies the live accumulators
            __builtin_amdgcn_sched_barrier(0);
        }
        // plane xy [iy][ix][c]: complete (mean over all 40 iz)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int iy = iy0 + wave * 4 + 2 * g + hi;
            plane_xy[((size_t)iy * RES + ix) * CD + n] = (TOut)(sum_z[g] * inv);
        }
        // plane xz [iz][ix][c]: fixed-order sum over this half's 5 strips
        __syncthreads();
        for (int i = tid; i < 40 * 32; i += blockDim.x) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 5; ++w) s += red[w * 40 * 32 + i];
            const int iz = i >> 5, c = i & 31;
            part_xz[((size_t)iz * RES + ix) * CD + c] = s;
        }
    }
    // plane yz partial [iz][iy][c] for this slab
    float* part = yz_partial + ((size_t)slab * B + b) * img_stride;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int iz = t * 8 + (r & 7), iy = iy0 + wave * 4 + 2 * (r >> 3) + hi;
            part[((size_t)iz * RES + iy) * CD + n] = acc_yz[t][r];
        }
}

template <typename TOut>
__global__ void plane_finalize_kernel(const float* __restrict__ xz_partial, const float* __restrict__ yz_partial,
                                      TOut* __restrict__ planes, int B, int nslab) {
    const size_t per = (size_t)B * RES * RES * CD;          // elements of one plane over the batch
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    planes[i] = (TOut)((xz_partial[i] + xz_partial[per + i]) * (1.0f / RES));
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += yz_partial[(size_t)k * per + i];
    planes[2 * per + i] = (TOut)(s * (1.0f / RES));
}

// ====================================================================================================
// Generic LDS-tiled implicit-GEMM convolution on MFMA, NHWC activations.
//   D[pixel][cout] = sum_{tap,cin} X[pixel+tap][cin] * W[cout][cin][tap]
//   A operand = pixels (32 per MFMA tile = 8 2x2 quads, so a lane's 16 D registers are 4 complete
//   quads and the fused 2x2 max-pool is in-lane), B operand = packed weight fragments from L2.
//   Workgroup = one (image, row strip) x one group of NB*32 output channels; input channels are
//   staged through LDS in chunks of 32 with a 16-byte pad per pixel (conflict-free ds_read_b128).
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
    int nimg;
};

template <typename T> struct Prec;
template <> struct Prec<float> { static constexpr int KG = 8; };      // channels per 16-byte k-group
template <> struct Prec<half_t> { static constexpr int KG = 16; };

template <typename T, int KIND, int C0, int C1, int COUT, int H, int W, int ROWS, int NW, int NB, bool POOL>
__global__ __launch_bounds__(NW * 64) void conv_mfma_kernel(ConvArgs a) {
    constexpr int CIN = C0 + C1;
    constexpr int TAPS = KIND == CONV3 ? 9 : 1;
    constexpr int HALO = KIND == CONV3 ? 1 : 0;
    constexpr int KG = Prec<T>::KG;                   // channels per k-group
    constexpr int KGC = 32 / KG;                      // k-groups per 32-channel chunk
    constexpr int NCHUNK = CIN / 32;
    constexpr int PS = 32 * (int)sizeof(T) + 16;      // LDS pixel stride in bytes
    constexpr int LW = W + 2 * HALO, LH = ROWS + 2 * HALO;
    constexpr int STRIPS = H / ROWS;
    constexpr int QUADS = (ROWS / 2) * (W / 2);
    constexpr int TILES = (QUADS + 7) / 8;
    constexpr int MT = (TILES + NW - 1) / NW;         // tiles per wave
    constexpr int NSUB = KIND == UPCONV ? 4 : 1;
    constexpr int NBT = COUT / 32;                    // 32-channel blocks per sub-output
    static_assert(H % ROWS == 0 && ROWS % 2 == 0 && W % 2 == 0, "strip geometry");
    static_assert(COUT % (32 * NB) == 0, "cout grouping");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int img = blockIdx.x / STRIPS, strip = blockIdx.x % STRIPS;
    const int y0 = strip * ROWS;
    // blockIdx.y enumerates (sub, cout group)
    const int sub = blockIdx.y / (NBT / NB), nb0 = (blockIdx.y % (NBT / NB)) * NB;

    // per-tile A geometry: M-row i = lane&31 : quad = i>>2, dy = (i>>1)&1, dx = i&1
    int a_off[MT];
    bool t_ok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int tile = wave + m * NW;
        const int q = tile * 8 + (n >> 2);
        t_ok[m] = tile < TILES;
        const int qq = q < QUADS ? q : QUADS - 1;
        const int y = 2 * (qq / (W / 2)) + ((n >> 1) & 1), x = 2 * (qq % (W / 2)) + (n & 1);
        a_off[m] = (y * LW + x) * PS + hi * 16;       // tap (0,0) of the haloed tile == pixel (y-1,x-1)
    }

    f32x16 acc[MT][NB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    const uint4* wfrag = reinterpret_cast<const uint4*>(a.w);
    constexpr int KGT = CIN / KG;                     // k-groups over all input channels

    for (int cc = 0; cc < NCHUNK; ++cc) {
        // ---- stage chunk cc (32 channels) of the haloed strip into LDS --------------------------
        const T* src = reinterpret_cast<const T*>(cc * 32 < C0 ? a.in0 : a.in1);
        const int csrc = cc * 32 < C0 ? C0 : C1;
        const int coff = cc * 32 < C0 ? cc * 32 : cc * 32 - C0;
        constexpr int VPP = 32 * (int)sizeof(T) / 16;           // 16-byte vectors per pixel
        __syncthreads();
        for (int i = tid; i < LH * LW * VPP; i += NW * 64) {
            const int pix = i / VPP, v = i % VPP;
            const int ly = pix / LW, lx = pix % LW;
            const int gy = y0 + ly - HALO, gx = lx - HALO;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                val = *reinterpret_cast<const uint4*>(src + ((size_t)(img * H + gy) * W + gx) * csrc + coff +
                                                      v * (16 / (int)sizeof(T)));
            *reinterpret_cast<uint4*>(smem + pix * PS + v * 16) = val;
        }
        __syncthreads();
        // ---- MFMA over taps x k-groups of this chunk --------------------------------------------
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = ((tap / 3) * LW + (tap % 3)) * PS;
#pragma unroll
            for (int kg = 0; kg < KGC; ++kg) {
                uint4 bw[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const size_t f = ((size_t)(sub * NBT + nb0 + j) * TAPS + tap) * KGT + cc * KGC + kg;
                    bw[j] = wfrag[f * 64 + lane];
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const uint4 av = *reinterpret_cast<const uint4*>(smem + a_off[m] + toff + kg * 32);
                    if constexpr (sizeof(T) == 2) {
                        const half8 A = __builtin_bit_cast(half8, av);
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[m][j] = mfma16(A, __builtin_bit_cast(half8, bw[j]), acc[m][j]);
                    } else {
                        const f32x4 A = __builtin_bit_cast(f32x4, av);
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const f32x4 Bv = __builtin_bit_cast(f32x4, bw[j]);
                            acc[m][j] = mfma32(A[0], Bv[0], acc[m][j]);
                            acc[m][j] = mfma32(A[1], Bv[1], acc[m][j]);
                            acc[m][j] = mfma32(A[2], Bv[2], acc[m][j]);
                            acc[m][j] = mfma32(A[3], Bv[3], acc[m][j]);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: bias (+ReLU) and stores.  Lane holds cout = nb*32 + n for 4 quads x 4 pixels ------
    T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!t_ok[m]) continue;
        const int tile = wave + m * NW;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co = (nb0 + j) * 32 + n;
            const float bv = a.bias[co];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                 // quad index within the tile = 2*g + hi
                const int q = tile * 8 + 2 * g + hi;
                if (q >= QUADS) continue;
                const int qy = q / (W / 2), qx = q % (W / 2);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[m][j][4 * g + e] + bv;
                    if (KIND == CONV3) v[e] = relu(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int y = y0 + 2 * qy + (e >> 1), x = 2 * qx + (e & 1);
                    if (KIND == UPCONV) {
                        const int oy = 2 * y + (sub >> 1), ox = 2 * x + (sub & 1);
                        out[((size_t)(img * 2 * H + oy) * (2 * W) + ox) * COUT + co] = (T)v[e];
                    } else {
                        out[((size_t)(img * H + y) * W + x) * COUT + co] = (T)v[e];
                        if (KIND == CONV1 && a.out_nchw)
                            a.out_nchw[((size_t)img * COUT + co) * H * W + y * W + x] = v[e];
                    }
                }
                if (POOL) {
                    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    T* op = reinterpret_cast<T*>(a.out_pool);
                    op[((size_t)(img * (H / 2) + y0 / 2 + qy) * (W / 2) + qx) * COUT + co] = (T)mx;
                }
            }
        }
    }
}

template <typename T, int KIND, int C0, int C1, int COUT, int H, int W, int ROWS, int NW, int NB, bool POOL>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr int HALO = KIND == CONV3 ? 1 : 0;
    constexpr int PS = 32 * (int)sizeof(T) + 16;
    constexpr size_t lds = (size_t)(ROWS + 2 * HALO) * (W + 2 * HALO) * PS;
    constexpr int NSUB = KIND == UPCONV ? 4 : 1;
    auto kern = conv_mfma_kernel<T, KIND, C0, C1, COUT, H, W, ROWS, NW, NB, POOL>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(a.nimg * (H / ROWS), NSUB * (COUT / 32 / NB));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

// ----------------------------------------------------------------------------------------------------
// Encoder driver.  Workspace carve (all NHWC, element type T):
//   P0 planes_in [3B,40,40,32] | A0 | S0 (skip 0) | Q0 [3B,20,20,32] | A1 [..,20,20,64] | S1 | Q1 [..,10,10,64]
//   | A2 [..,10,10,128] | S2 | U0 [..,20,20,64] | A3 | A4 | U1 [..,40,40,32] | A5 | A6 | YZ, XZ partials fp32
// ----------------------------------------------------------------------------------------------------
struct EncWs {
    size_t P0, A0, S0, Q0, A1, S1, Q1, A2, S2, U0, A3, A4, U1, A5, A6, YZ, XZ, total;
};
EncWs enc_workspace(int B, int precision, int nslab) {
    const size_t es = precision == 1 ? 2 : 4;
    const size_t n = 3 * (size_t)B;
    EncWs w{};
    size_t at = 0;
    auto take = [&](size_t elems, size_t esz) { size_t o = at; at += align_up(elems * esz, 256); return o; };
    w.P0 = take(n * 1600 * 32, es); w.A0 = take(n * 1600 * 32, es); w.S0 = take(n * 1600 * 32, es);
    w.Q0 = take(n * 400 * 32, es);  w.A1 = take(n * 400 * 64, es);  w.S1 = take(n * 400 * 64, es);
    w.Q1 = take(n * 100 * 64, es);  w.A2 = take(n * 100 * 128, es); w.S2 = take(n * 100 * 128, es);
    w.U0 = take(n * 400 * 64, es);  w.A3 = take(n * 400 * 64, es);  w.A4 = take(n * 400 * 64, es);
    w.U1 = take(n * 1600 * 32, es); w.A5 = take(n * 1600 * 32, es); w.A6 = take(n * 1600 * 32, es);
    w.YZ = take((size_t)nslab * B * 1600 * 32, 4);
    w.XZ = take((size_t)2 * B * 1600 * 32, 4);
    w.total = at;
    return w;
}

int enc_nslab(int B) { return B >= 8 ? 5 : B >= 4 ? 10 : 20; }

// probe: if probe_stage == k, ev0/ev1 (hipEvent_t) are recorded right before / after launch k
// (k = 0 conv_in+project, 1 plane_finalize, 2..14 = U-Net layers 0..12).
struct Probe { int stage; hipEvent_t ev0, ev1; };

template <typename T>
static int encoder_run(const float* tsdf, const uint8_t* blob, void* planes_nhwc, float* planes_nchw, int B,
                       uint8_t* ws, hipStream_t s, const Probe& pr) {
    int stage_no = 0;
    auto pre = [&]() { if (pr.stage == stage_no) (void)hipEventRecord(pr.ev0, s); };
    auto post = [&]() { if (pr.stage == stage_no) (void)hipEventRecord(pr.ev1, s); ++stage_no; };
    constexpr int precision = sizeof(T) == 2 ? 1 : 0;
    const PackOff ko = pack_offsets();
    const int nslab = enc_nslab(B);
    const EncWs w = enc_workspace(B, precision, nslab);
    T* P0 = reinterpret_cast<T*>(ws + w.P0);
    float* YZ = reinterpret_cast<float*>(ws + w.YZ);
    float* XZ = reinterpret_cast<float*>(ws + w.XZ);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convin_project_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)CI_LDS_BYTES);
    pre();
    hipLaunchKernelGGL(convin_project_kernel<T>, dim3(nslab, B, 2), dim3(320), CI_LDS_BYTES, s, tsdf,
                       reinterpret_cast<const float*>(blob + ko.convin_w),
                       reinterpret_cast<const float*>(blob + ko.convin_b), P0, XZ, YZ, B, RES / nslab);
    post();
    {
        pre();
        const size_t per = (size_t)B * RES * RES * CD;
        hipLaunchKernelGGL(plane_finalize_kernel<T>, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, XZ, YZ,
                           P0, B, nslab);
        post();
    }
    if (hipGetLastError() != hipSuccess) return -10;

    const int nimg = 3 * B;
    auto W_ = [&](int l) { return blob + (precision == 1 ? ko.conv[l].w16 : ko.conv[l].w32); };
    auto Bi = [&](int l) { return reinterpret_cast<const float*>(blob + ko.conv[l].bias); };
    auto args = [&](int l, const void* i0, const void* i1, void* o, void* op) {
        ConvArgs a{};
        a.in0 = i0; a.in1 = i1; a.w = W_(l); a.bias = Bi(l); a.out = o; a.out_pool = op; a.out_nchw = nullptr;
        a.nimg = nimg;
        return a;
    };
    uint8_t* b = ws;
    int rc = 0;
    // template params: <T, KIND, C0, C1, COUT, H, W, ROWS, NW, NB, POOL>
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 32, 40, 40, 8, 5, 1, false>(args(0, b + w.P0, nullptr, b + w.A0, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 32, 40, 40, 8, 5, 1, true>(args(1, b + w.A0, nullptr, b + w.S0, b + w.Q0), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 64, 20, 20, 10, 4, 1, false>(args(2, b + w.Q0, nullptr, b + w.A1, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 0, 64, 20, 20, 10, 4, 1, true>(args(3, b + w.A1, nullptr, b + w.S1, b + w.Q1), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 0, 128, 10, 10, 10, 4, 1, false>(args(4, b + w.Q1, nullptr, b + w.A2, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 128, 0, 128, 10, 10, 10, 4, 1, false>(args(5, b + w.A2, nullptr, b + w.S2, nullptr), s); post();
    pre(); rc |= launch_conv<T, UPCONV, 128, 0, 64, 10, 10, 10, 4, 2, false>(args(6, b + w.S2, nullptr, b + w.U0, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 64, 64, 20, 20, 10, 4, 1, false>(args(7, b + w.U0, b + w.S1, b + w.A3, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 0, 64, 20, 20, 10, 4, 1, false>(args(8, b + w.A3, nullptr, b + w.A4, nullptr), s); post();
    pre(); rc |= launch_conv<T, UPCONV, 64, 0, 32, 20, 20, 10, 4, 1, false>(args(9, b + w.A4, nullptr, b + w.U1, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 32, 32, 40, 40, 8, 5, 1, false>(args(10, b + w.U1, b + w.S0, b + w.A5, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 32, 40, 40, 8, 5, 1, false>(args(11, b + w.A5, nullptr, b + w.A6, nullptr), s); post();
    {
        ConvArgs a = args(12, b + w.A6, nullptr, planes_nhwc, nullptr);
        a.out_nchw = planes_nchw;
        pre(); rc |= launch_conv<T, CONV1, 32, 0, 32, 40, 40, 8, 5, 1, false>(a, s); post();
    }
    return rc;
}

int launch_encoder(const float* tsdf, const uint8_t* blob, void* planes_nhwc, float* planes_nchw, int B,
                   int precision, uint8_t* ws, hipStream_t s, int probe_stage, void* ev0, void* ev1) {
    if (B <= 0) return 0;
    Probe pr{ev0 && ev1 ? probe_stage : -1, static_cast<hipEvent_t>(ev0), static_cast<hipEvent_t>(ev1)};
    return precision == 1 ? encoder_run<half_t>(tsdf, blob, planes_nhwc, planes_nchw, B, ws, s, pr)
                          : encoder_run<float>(tsdf, blob, planes_nhwc, planes_nchw, B, ws, s, pr);
}

}  // namespace giga
