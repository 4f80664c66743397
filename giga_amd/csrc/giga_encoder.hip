// Encoder kernels: TSDF (B,40,40,40) -> three 40x40x32 feature planes (NHWC).
//
// Replaces reference LocalVoxelEncoder.forward (ConvONets/encoder/voxels.py:89-121):
//   relu(Conv3d(1,32,3,pad=1)) -> permute -> 3x [normalize_coordinate, coordinate2index,
//   torch_scatter.scatter_mean] -> shared UNet (encoder/unet.py:225-239).
// Kernel 1 (convin_project_kernel) fuses the 3-D conv, the ReLU and the three axis means, so the
// 32x40^3 feature volume (8.2 MB/scene in the reference) is never written to memory.  Kernel 2
// (conv16_kernel) is one LDS-staged implicit-GEMM convolution used for every U-Net layer.
#include "giga_dev.h"
#include "giga_conv16.h"

namespace giga {

// ====================================================================================================
// conv_in + ReLU + axis means.
//   grid (NSLAB, B), block 512 = 8 waves = 4 iy-groups (10 rows each) x 2 channel halves.  The
//   workgroup walks SX = 40/NSLAB consecutive ix slices of one scene.  Work unit = 16 voxels
//   (2 iy x 8 iz) x 16 channels: D[voxel][channel] = A[voxel][tap] * Wt[tap][channel], K = 27 taps
//   (+1 zero) = 7 x v_mfma_f32_16x16x4_f32 (exact fp32).  25 units per wave per slice on every one of
//   the 8 waves: the 4 SIMDs of the CU carry exactly the same MFMA load (2 waves each).
//   D-row -> voxel map: row v = 4g + r (g = lane>>4):  iz_local = 4*(g&1) + r ,  iy_local = g>>1, so
//   * mean over iz (plane 'xy') : in-lane adds + one lane^16 exchange               -> written directly
//   * mean over iy (plane 'xz') : in-lane adds + one lane^32 exchange, then a fixed-order sum of the
//                                 4 iy-groups through LDS                            -> written directly
//   * mean over ix (plane 'yz') : register accumulation over the slab, one fp32 partial per slab to
//                                 HBM, summed in fixed order by plane_finalize_kernel (no atomics).
// Plane pixel (H,W) conventions (common.py:246-251,303-318): xz -> [iz][ix], xy -> [iy][ix], yz -> [iz][iy].
// ====================================================================================================

constexpr int CI_ROWSTRIDE = 56;                  // floats per LDS row (>= 42; 56 mod 32 = 24 spreads the rows over banks)
constexpr int CI_SLICE = 42 * CI_ROWSTRIDE;       // one haloed slice
constexpr int CI_LDS_SLICES = 4 * CI_SLICE;       // ring of 4 slices (floats): 3 in use + 1 being filled
constexpr int CI_LDS_RED = 4 * 40 * 32;           // cross-group reduction buffer (floats), double-buffered
constexpr size_t CI_LDS_BYTES = (CI_LDS_SLICES + 2 * CI_LDS_RED) * sizeof(float);

template <typename TOut>
__global__ __launch_bounds__(512) void convin_project_kernel(
    const float* __restrict__ tsdf,        // [B][40][40][40]
    const float* __restrict__ wpk,         // [2][7][64] packed B operands
    const float* __restrict__ bias,        // [32]
    TOut* __restrict__ planes,             // [3][B][40][40][32] NHWC (xz, xy written here)
    float* __restrict__ yz_partial,        // [NSLAB][B][40(iz)][40(iy)][32] sums over SX ix
    int B, int SX) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* slices = lds;
    float* red = lds + CI_LDS_SLICES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int chh = wave & 1, grp = wave >> 1;
    const int slab = blockIdx.x, b = blockIdx.y;
    const int ix0 = slab * SX;
    const float* vol = tsdf + (size_t)b * RES * RES * RES;

    for (int i = tid; i < CI_LDS_SLICES; i += blockDim.x) slices[i] = 0.f;   // halos stay zero
    float wreg[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) wreg[s] = wpk[(chh * 7 + s) * 64 + lane];
    const float bn = bias[16 * chh + j];
    __syncthreads();

    // slice ix lives in ring slot (ix+1) & 3; out-of-range slices are zero.  Each thread moves up to 4
    // voxels of a slice (1600 = 3*512 + 64): global -> registers early, registers -> LDS late.
    float pre[4];
    auto fetch_slice = [&](int ix) {
        const bool in = ix >= 0 && ix < RES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q;
            pre[q] = (in && i < RES * RES) ? vol[(size_t)ix * RES * RES + i] : 0.f;
        }
    };
    auto commit_slice = [&](int ix) {
        float* dst = slices + ((ix + 1) & 3) * CI_SLICE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q;         // recomputed (not kept live): the kernel sits at the VGPR limit
            if (i < RES * RES) dst[(i / RES + 1) * CI_ROWSTRIDE + (i % RES + 1)] = pre[q];
        }
    };
    fetch_slice(ix0 - 1); commit_slice(ix0 - 1);
    fetch_slice(ix0);     commit_slice(ix0);
    fetch_slice(ix0 + 1); commit_slice(ix0 + 1);
    __syncthreads();

    // A operand: row i = lane&15 is voxel (iy_l = i>>3, iz_l = 4*((i>>2)&1) + (i&3)); k-slot g supplies tap 4s+g
    const int a_base = (grp * 10 + (j >> 3)) * CI_ROWSTRIDE + 4 * ((j >> 2) & 1) + (j & 3);
    f32x4v acc_yz[5][5];
#pragma unroll
    for (int ip = 0; ip < 5; ++ip)
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) acc_yz[ip][zg] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const float inv = 1.0f / RES;
    const size_t img_stride = (size_t)RES * RES * CD;
    TOut* plane_xz = planes + ((size_t)0 * B + b) * img_stride;
    TOut* plane_xy = planes + ((size_t)1 * B + b) * img_stride;
    const int ch = 16 * chh + j;

    for (int sx = 0; sx < SX; ++sx) {
        const int ix = ix0 + sx;
        // Hazards are covered by the single barrier below: the slot filled this iteration (slice ix+2)
        // was last read as slice ix-2 in the previous iteration; `red` alternates between two buffers.
        float* redw = red + (sx & 1) * CI_LDS_RED;
        if (sx + 1 < SX) fetch_slice(ix + 2);        // global loads fly under this slice's MFMAs
        const int o0 = ((ix + 0) & 3) * CI_SLICE, o1 = ((ix + 1) & 3) * CI_SLICE, o2 = ((ix + 2) & 3) * CI_SLICE;
        // k-slot g supplies tap 4s+g (tap = dx*9 + dy*3 + dz, tap 27 has zero weight).  Recomputed per
        // slice from an opaque copy of g so the 14 per-lane constants are not kept live (VGPR limit).
        int gq = g;
        asm volatile("" : "+v"(gq));
        int aoff[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            int t = 4 * s + gq;
            t = t > 26 ? 26 : t;
            const int dx = t / 9;
            aoff[s] = a_base + ((t / 3) % 3) * CI_ROWSTRIDE + t % 3 + (dx == 0 ? o0 : dx == 1 ? o1 : o2);
        }
        float sum_z[5];                   // per iy-pair: sum over the 5 iz-groups and the 4 in-lane iz
#pragma unroll
        for (int ip = 0; ip < 5; ++ip) sum_z[ip] = 0.f;
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) {
            float part_y[4] = {0.f, 0.f, 0.f, 0.f};   // per r: sum over the 5 iy-pairs of this group
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) {
                f32x4v d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 7; ++s)
                    d = mfma32_16(slices[aoff[s] + 2 * ip * CI_ROWSTRIDE + 8 * zg], wreg[s], d);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = relu(d[r] + bn);
                    acc_yz[ip][zg][r] += v;
                    sum_z[ip] += v;
                    part_y[r] += v;
                }
                // two independent 7-MFMA chains in flight cover the 40-cycle latency of the 32-cycle
                // 16x16x4 MFMA; more only multiplies the live A operands (the kernel is VGPR-bound)
                if (ip & 1) __builtin_amdgcn_sched_barrier(0);
            }
            // plane xz [iz][ix][c]: the other iy row of each tile lives in lane^32; 4 groups go through LDS
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = part_y[r] + __shfl_xor(part_y[r], 32);
                if ((g >> 1) == 0) redw[(grp * 40 + 8 * zg + 4 * (g & 1) + r) * 32 + ch] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // plane xy [iy][ix][c]: other half of the 8 iz of each tile lives in lane^16
#pragma unroll
        for (int ip = 0; ip < 5; ++ip) {
            const float s = sum_z[ip] + __shfl_xor(sum_z[ip], 16);
            if ((g & 1) == 0) {
                const int iy = grp * 10 + 2 * ip + (g >> 1);
                plane_xy[((size_t)iy * RES + ix) * CD + ch] = (TOut)(s * inv);
            }
        }
        if (sx + 1 < SX) commit_slice(ix + 2);
        __syncthreads();
        // the fixed-order sum of the 4 iy-groups is done by ONE half of the waves (alternating per slice):
        // the sibling wave of every SIMD goes straight on to the next slice's MFMAs
        if ((wave >> 2) == (sx & 1)) {
            for (int i = tid & 255; i < 40 * 32; i += 256) {
                const float s = (redw[i] + redw[1280 + i]) + (redw[2560 + i] + redw[3840 + i]);
                const int iz = i >> 5, c = i & 31;
                plane_xz[((size_t)iz * RES + ix) * CD + c] = (TOut)(s * inv);
            }
        }
    }
    // plane yz partial [iz][iy][c] for this slab
    float* part = yz_partial + ((size_t)slab * B + b) * img_stride;
#pragma unroll
    for (int ip = 0; ip < 5; ++ip)
#pragma unroll
        for (int zg = 0; zg < 5; ++zg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iz = 8 * zg + 4 * (g & 1) + r, iy = grp * 10 + 2 * ip + (g >> 1);
                part[((size_t)iz * RES + iy) * CD + ch] = acc_yz[ip][zg][r];
            }
}

template <typename TOut>
__global__ void plane_finalize_kernel(const float* __restrict__ yz_partial, TOut* __restrict__ planes, int B,
                                      int nslab) {
    const size_t per = (size_t)B * RES * RES * CD;          // elements of one plane over the batch
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += yz_partial[(size_t)k * per + i];
    planes[2 * per + i] = (TOut)(s * (1.0f / RES));
}

// ----------------------------------------------------------------------------------------------------
// Encoder driver.  Workspace carve (all NHWC, element type T):
//   P0 planes_in [3B,40,40,32] | A0 | S0 (skip 0) | Q0 [3B,20,20,32] | A1 [..,20,20,64] | S1 | Q1 [..,10,10,64]
//   | A2 [..,10,10,128] | S2 | U0 [..,20,20,64] | A3 | A4 | U1 [..,40,40,32] | A5 | A6 | YZ partials fp32
// ----------------------------------------------------------------------------------------------------
struct EncWs {
    size_t P0, A0, S0, Q0, A1, S1, Q1, A2, S2, U0, A3, A4, U1, A5, A6, YZ, XZ, total;   // XZ unused (kept for the ABI)
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
    w.XZ = w.YZ;
    w.total = at;
    return w;
}

// slabs per scene: the smallest divisor of 40 that gives >= 256 workgroups (one per CU), at most 40
int enc_nslab(int B) {
    const int divs[8] = {1, 2, 4, 5, 8, 10, 20, 40};
    for (int d : divs)
        if ((long long)B * d >= 256) return d;
    return 40;
}

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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convin_project_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)CI_LDS_BYTES);
    pre();
    hipLaunchKernelGGL(convin_project_kernel<T>, dim3(nslab, B), dim3(512), CI_LDS_BYTES, s, tsdf,
                       reinterpret_cast<const float*>(blob + ko.convin_w),
                       reinterpret_cast<const float*>(blob + ko.convin_b), P0, YZ, B, RES / nslab);
    post();
    {
        pre();
        const size_t per = (size_t)B * RES * RES * CD;
        hipLaunchKernelGGL(plane_finalize_kernel<T>, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, YZ, P0, B,
                           nslab);
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
    // template params: <T, KIND, C0, C1, COUT, H, W, NB (16-channel blocks per unit), POOL>
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 32, 40, 40, 2, false>(args(0, b + w.P0, nullptr, b + w.A0, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 32, 40, 40, 2, true>(args(1, b + w.A0, nullptr, b + w.S0, b + w.Q0), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 64, 20, 20, 1, false>(args(2, b + w.Q0, nullptr, b + w.A1, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 0, 64, 20, 20, 1, true>(args(3, b + w.A1, nullptr, b + w.S1, b + w.Q1), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 0, 128, 10, 10, 1, false>(args(4, b + w.Q1, nullptr, b + w.A2, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 128, 0, 128, 10, 10, 1, false>(args(5, b + w.A2, nullptr, b + w.S2, nullptr), s); post();
    pre(); rc |= launch_conv<T, UPCONV, 128, 0, 64, 10, 10, 2, false>(args(6, b + w.S2, nullptr, b + w.U0, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 64, 64, 20, 20, 1, false>(args(7, b + w.U0, b + w.S1, b + w.A3, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 64, 0, 64, 20, 20, 1, false>(args(8, b + w.A3, nullptr, b + w.A4, nullptr), s); post();
    pre(); rc |= launch_conv<T, UPCONV, 64, 0, 32, 20, 20, 2, false>(args(9, b + w.A4, nullptr, b + w.U1, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 32, 32, 40, 40, 2, false>(args(10, b + w.U1, b + w.S0, b + w.A5, nullptr), s); post();
    pre(); rc |= launch_conv<T, CONV3, 32, 0, 32, 40, 40, 2, false>(args(11, b + w.A5, nullptr, b + w.A6, nullptr), s); post();
    {
        ConvArgs a = args(12, b + w.A6, nullptr, planes_nhwc, nullptr);
        a.out_nchw = planes_nchw;
        pre(); rc |= launch_conv<T, CONV1, 32, 0, 32, 40, 40, 2, false>(a, s); post();
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
