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

#ifdef CI_TRACE   // diagnostic build: per-wave issue timeline of workgroup (0,0) into the yz partial buffer
#define CI_T(idx) do { if (slab == 0 && b == 0 && lane == 0) reinterpret_cast<long long*>(yz_partial)[wave * 128 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CI_T(idx) do {} while (0)
#endif

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
    auto fetch_slice = [&](int ix, float (&pre)[4]) {
        const bool in = ix >= 0 && ix < RES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q;
            pre[q] = (in && i < RES * RES) ? vol[(size_t)ix * RES * RES + i] : 0.f;
        }
    };
    auto commit_slice = [&](int ix, const float (&pre)[4]) {
        float* dst = slices + ((ix + 1) & 3) * CI_SLICE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q;         // recomputed (not kept live): the kernel sits at the VGPR limit
            if (i < RES * RES) dst[(i / RES + 1) * CI_ROWSTRIDE + (i % RES + 1)] = pre[q];
        }
    };
    {   // the three slices of the first step: one round trip, not three
        float p0[4], p1[4], p2[4];
        fetch_slice(ix0 - 1, p0); fetch_slice(ix0, p1); fetch_slice(ix0 + 1, p2);
        commit_slice(ix0 - 1, p0); commit_slice(ix0, p1); commit_slice(ix0 + 1, p2);
    }
    __syncthreads();

    // A operand: row i = lane&15 is voxel (iy_l = i>>3, iz_l = 4*((i>>2)&1) + (i&3)); k-slot g supplies tap 4s+g
    // (tap = dx*9 + dy*3 + dz; tap 27 has zero weight and reads tap 26's voxel).  abase[s] is loop-invariant;
    // only the ring slot of the tap's dx changes per slice.
    const int a_base = (grp * 10 + (j >> 3)) * CI_ROWSTRIDE + 4 * ((j >> 2) & 1) + (j & 3);
    int abase[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        int t = 4 * s + g;
        t = t > 26 ? 26 : t;
        abase[s] = a_base + ((t / 3) % 3) * CI_ROWSTRIDE + t % 3;
    }
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
    const f32x4v bias4 = {bn, bn, bn, bn};            // the bias rides in the C operand of the first MFMA
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int sx = 0; sx < SX; ++sx) {
        const int ix = ix0 + sx;
        // Hazards are covered by the single barrier below: the slot filled this iteration (slice ix+2)
        // was last read as slice ix-2 in the previous iteration; `red` alternates between two buffers.
        float* redw = red + (sx & 1) * CI_LDS_RED;
        float pre[4];
        if (sx + 1 < SX) fetch_slice(ix + 2, pre);   // global loads fly under this slice's MFMAs
        const int o0 = ((ix + 0) & 3) * CI_SLICE, o1 = ((ix + 1) & 3) * CI_SLICE, o2 = ((ix + 2) & 3) * CI_SLICE;
        int addr[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int lo = (4 * s) / 9, hi = (4 * s + 3 > 26 ? 26 : 4 * s + 3) / 9;    // dx of k-slots 0 and 3
            const int olo = lo == 0 ? o0 : lo == 1 ? o1 : o2, ohi = hi == 0 ? o0 : hi == 1 ? o1 : o2;
            addr[s] = abase[s] + (lo == hi ? olo : (4 * s + g >= 9 * hi ? ohi : olo));
        }
        f32x2v sum_z[5];                  // per iy-pair: sum over the 5 iz-groups and the 4 in-lane iz (two partials)
#pragma unroll
        for (int ip = 0; ip < 5; ++ip) sum_z[ip] = f32x2v{0.f, 0.f};
        // Software pipeline over the 5 iz-groups.  Per group: five independent 7-MFMA chains (the iy-pairs).
        // A operands of MFMA steps 0..2 are loaded one group ahead (P), those of steps 3..6 at the top of the
        // group (Q) under the first 15 MFMAs, so no LDS round trip is exposed.  The 35 MFMAs stay one
        // uninterrupted burst: an extra issue slot between MFMAs costs far more than the slot itself, and the
        // ReLU / axis-sum epilogue (packed adds) runs as its own burst under the sibling wave's MFMAs.
        float P[3][5], Q[4][5];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) P[s][ip] = slices[addr[s] + 2 * ip * CI_ROWSTRIDE];
        CI_T(sx * 16 + 0);
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) {
            f32x4v d[5];
#pragma unroll
            for (int s = 3; s < 7; ++s)
#pragma unroll
                for (int ip = 0; ip < 5; ++ip) Q[s - 3][ip] = slices[addr[s] + 2 * ip * CI_ROWSTRIDE + 8 * zg];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int ip = 0; ip < 5; ++ip) d[ip] = mfma32_16(P[s][ip], wreg[s], s == 0 ? bias4 : d[ip]);
            __builtin_amdgcn_sched_barrier(0);
            if (zg + 1 < 5) {
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int ip = 0; ip < 5; ++ip)
                        P[s][ip] = slices[addr[s] + 2 * ip * CI_ROWSTRIDE + 8 * (zg + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 3; s < 7; ++s)
#pragma unroll
                for (int ip = 0; ip < 5; ++ip) d[ip] = mfma32_16(Q[s - 3][ip], wreg[s], d[ip]);
            __builtin_amdgcn_sched_barrier(0);
            CI_T(sx * 16 + 1 + 2 * zg);
            f32x4v part_y = zero4;            // per r: sum over the 5 iy-pairs of this group
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) {
                const f32x4v v = __builtin_elementwise_max(d[ip], zero4);
                acc_yz[ip][zg] += v;
                sum_z[ip] += f32x2v{v[0], v[1]} + f32x2v{v[2], v[3]};
                part_y += v;
            }
            // plane xz [iz][ix][c]: the other iy row of each tile lives in lane^32; 4 groups go through LDS
            float other[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) other[r] = __shfl_xor(part_y[r], 32);       // four exchanges, one wait
            if ((g >> 1) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    redw[(grp * 40 + 8 * zg + 4 * (g & 1) + r) * 32 + ch] = part_y[r] + other[r];
            }
            __builtin_amdgcn_sched_barrier(0);
            CI_T(sx * 16 + 2 + 2 * zg);
        }
        // plane xy [iy][ix][c]: other half of the 8 iz of each tile lives in lane^16
#pragma unroll
        for (int ip = 0; ip < 5; ++ip) {
            const float sl = sum_z[ip][0] + sum_z[ip][1];
            const float sm = sl + __shfl_xor(sl, 16);
            if ((g & 1) == 0) {
                const int iy = grp * 10 + 2 * ip + (g >> 1);
                plane_xy[((size_t)iy * RES + ix) * CD + ch] = (TOut)(sm * inv);
            }
        }
        if (sx + 1 < SX) commit_slice(ix + 2, pre);
        CI_T(sx * 16 + 11);
        __syncthreads();
        CI_T(sx * 16 + 12);
        // the fixed-order sum of the 4 iy-groups is done by ONE half of the waves (alternating per slice):
        // the sibling wave of every SIMD goes straight on to the next slice's MFMAs
        if ((wave >> 2) == (sx & 1)) {
            for (int i = tid & 255; i < 40 * 32; i += 256) {
                const float sr = (redw[i] + redw[1280 + i]) + (redw[2560 + i] + redw[3840 + i]);
                const int iz = i >> 5, c = i & 31;
                plane_xz[((size_t)iz * RES + ix) * CD + c] = (TOut)(sr * inv);
            }
        }
        CI_T(sx * 16 + 13);
    }
#ifdef CI_TRACE
    return;
#endif
    // plane yz partial of this slab, in the kernel's own register order so that every store is one fully
    // coalesced 16 B per lane: [slab][b][wave][unit = ip*5+zg][lane][r]; plane_finalize_kernel undoes the map.
    f32x4v* part = reinterpret_cast<f32x4v*>(yz_partial + ((size_t)slab * B + b) * img_stride) + (size_t)wave * 25 * 64 + lane;
#pragma unroll
    for (int ip = 0; ip < 5; ++ip)
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) part[(ip * 5 + zg) * 64] = acc_yz[ip][zg];
}

template <typename TOut>
__global__ void plane_finalize_kernel(const float* __restrict__ yz_partial, TOut* __restrict__ planes, int B,
                                      int nslab) {
    // one thread per (scene, iy, group of 4 iz, channel): a 16-B read per slab in convin_project's register
    // order (see the store at its end), fixed-order sum over the slabs, four channel-contiguous row writes
    const size_t per = (size_t)B * RES * RES * CD;          // elements of one plane over the batch
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per / 4) return;
    const int c = (int)(t % CD);
    const int izq = (int)((t / CD) % 10);
    const int iy = (int)((t / (CD * 10)) % RES);
    const size_t b = t / ((size_t)CD * 10 * RES);
    const int iyl = iy % 10;
    const int wave = (iy / 10) * 2 + c / 16;
    const int unit = (iyl / 2) * 5 + izq / 2;
    const int lane = ((iyl & 1) * 2 + (izq & 1)) * 16 + (c & 15);
    const f32x4v* src = reinterpret_cast<const f32x4v*>(yz_partial) + (b * 8 + wave) * 25 * 64 + unit * 64 + lane;
    f32x4v sum = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nslab; ++k) sum += src[(size_t)k * (per / 4)];
    TOut* dst = planes + 2 * per + ((b * RES + 4 * izq) * RES + iy) * CD + c;      // [iz][iy][c]
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)r * RES * CD] = (TOut)(sum[r] * (1.0f / RES));
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
        hipLaunchKernelGGL(plane_finalize_kernel<T>, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, s, YZ, P0, B,
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
        a.trace_id = l;
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

#ifdef GIGA_TRACE
// diagnostic build only: select the traced U-Net layer (host_out == nullptr) or read the timeline back
extern "C" int giga_debug_conv_trace(int layer, long long* host_out) {
    if (!host_out) return hipMemcpyToSymbol(HIP_SYMBOL(giga::g_conv_trace_layer), &layer, sizeof(int)) == hipSuccess ? 0 : -10;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_conv_trace), sizeof(long long) * giga::CONV_NW * 64) == hipSuccess ? 0 : -10;
}
#endif
