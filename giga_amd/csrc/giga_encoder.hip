// Encoder kernels: TSDF (B,40,40,40) -> three 40x40x32 feature planes (NHWC).
//
// Replaces reference LocalVoxelEncoder.forward (ConvONets/encoder/voxels.py:89-121):
//   relu(Conv3d(1,32,3,pad=1)) -> permute -> 3x [normalize_coordinate, coordinate2index,
//   torch_scatter.scatter_mean] -> shared UNet (encoder/unet.py:225-239).
// Kernel 1 (convin_project_kernel) fuses the 3-D conv, the ReLU and the three axis means, so the
// 32x40^3 feature volume (8.2 MB/scene in the reference) is never written to memory.  Kernel 2
// (conv16_kernel) is one LDS-staged implicit-GEMM convolution used for every U-Net layer.
#include <atomic>
#include <cstdlib>

#include "../../include/giga_hip.h"
#include "giga_dev.h"
#include "giga_conv16.h"
#include "giga_conv32.h"
#include "giga_wino.h"
#include "giga_bwd_mega.h"
#include "giga_args.h"

namespace giga {


// ====================================================================================================
// conv_in + ReLU + axis means.
//   grid (NXP, 8, B), block 512 = 8 waves.  A workgroup owns one iy-group (10 rows) x one channel half (16 channels) of
//   one scene for XW = 8*SXW consecutive ix slices; it stages that haloed sub-volume ((XW+2) x 12 x 42 voxels, <= 87 KiB)
//   in LDS ONCE and every wave then walks its own SXW slices with no workgroup barrier at all.
//   Work unit = 16 voxels (2 iy x 8 iz) x 16 channels: D[voxel][channel] = A[voxel][tap] * Wt[tap][channel], K = 27 taps
//   (+1 zero) = 7 x v_mfma_f32_16x16x4_f32 (exact fp32), 25 units per slice.
//   D-row -> voxel map: row v = 4g + r (g = lane>>4):  iz_local = 4*(g&1) + r ,  iy_local = g>>1, so
//   * mean over iz (plane 'xy') : in-lane adds + one lane^16 exchange                          -> written directly
//   * mean over iy (plane 'xz') : in-lane adds + one lane^32 exchange = the sum over the group's 10 rows; one fp32
//                                 partial per iy-group, the 4 groups are summed by plane_finalize_kernel
//   * mean over ix (plane 'yz') : register accumulation over the wave's slices, then a fixed-order sum of the 8 waves
//                                 through LDS (and over the NXP x-parts in plane_finalize_kernel); no atomics anywhere.
//   SXW = 5 (NXP = 1, batch >= 32: 8*B workgroups) or 1 (NXP = 5: 40*B workgroups for small batches).
// Plane pixel (H,W) conventions (common.py:246-251,303-318): xz -> [iz][ix], xy -> [iy][ix], yz -> [iz][iy].
// ====================================================================================================

constexpr int CV_RS = 44;                         // floats per (ix, iy) row of the staged sub-volume
constexpr int CV_OFF = 3;                         // iz = -1 sits at float CV_OFF of its row, iz = 40 at float 44 = float 0 of the next row
constexpr int CV_ROWS = 12;                       // the group's 10 iy rows + halo
constexpr int ci_red_units(int nw) { return nw == 4 ? 25 : 13; }   // yz reduction: units per round (8 waves: 13 + 12, 104 KiB; 4 waves: all 25)
constexpr size_t ci_lds_bytes(int xw, int nw, bool f16class = false) {
    const size_t stage = ((size_t)(xw + 3) * (f16class ? CI16_SLAB : CV_ROWS * CV_RS) + 4) * sizeof(float);   // + one slab: the A-operand prefetch runs one slice ahead
    const size_t red = (size_t)nw * ci_red_units(nw) * 64 * 16;                // NW waves x UR units x 64 lanes x 16 B
    return stage > red ? stage : red;
}

// LO = false (plain f16 mode): only the hi x hi product, i.e. f16 operands / fp32 accumulate, one MFMA per unit.
// SPLIT: f16x3 split-operand arithmetic for the 27-tap contraction.  The sub-volume is staged as one 32-bit word per voxel
// holding the pair (hi = f16(v), lo = f16(v - hi)); the 27 taps (+5 zero weights) are ONE K = 32 step, i.e. three
// v_mfma_f32_16x16x32_f16 per unit (W_lo*x_hi + W_hi*x_lo + W_hi*x_hi, bias in the C operand) instead of seven
// v_mfma_f32_16x16x4_f32; a lane gathers its 8 tap words and separates them into the hi and lo operand with 8 byte-permutes.
#ifdef GIGA_TRACE
static __device__ long long g_ci_trace[8 * 64];
#define CI_T(idx) do { if (blockIdx.x == 43 && lane == 0) \
        g_ci_trace[wave * 64 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CI_T(idx) do {} while (0)
#endif
// MASK (training forward): the sign bits of the pre-activations go to `relu_mask` -- per (scene, workgroup row wy, ix) and lane one
// 16-byte word, value k = (zg * 5 + ip) * 4 + r of the lane's 100 in word k >> 5, shifted in from the right (v_alignbit: one VALU
// instruction per value) -- so that the backward (convin_bwd_kernel<.., true>) does not recompute the convolution for its ReLU mask.
// A set bit = negative (or -0): gradient 0; +0 counts as positive (measure zero).
template <typename TOut, int SXW, bool SPLIT = false, int NW = 8, bool LO = true, bool MASK = false>
__global__ __launch_bounds__(512) void convin_project_kernel(   // (512 also for NW = 4: a 256-thread bound makes the compiler put the MFMA results into AGPRs and copy them out for the epilogue)
    const float* __restrict__ tsdf,        // [B][40][40][40]
    const float* __restrict__ wpk,         // [2][7][64] packed B operands (SPLIT: [2][hi|lo][64] x 8 halfs)
    const float* __restrict__ bias,        // [32]
    TOut* __restrict__ planes,             // [3][B][40][40][32] NHWC (xy written here)
    float* __restrict__ xz_partial,        // [4 iy-groups][B][40(iz)][40(ix)][32] sums over the group's 10 iy
    float* __restrict__ yz_partial,        // [NXP][B][40(iz)][40(iy)][32] sums over the part's ix
    int B, uint4* __restrict__ relu_mask = nullptr) {
    constexpr int XW = NW * SXW, NT = NW * 64;
    // row / slab strides of the staged sub-volume in words: the f16-class instantiations use the layout that makes their 8-tap
    // gather conflict-free (giga_layout.h: ci16_tap), fp32 the compact one (its 4-tap k-steps are conflict-free there)
    constexpr int RS = SPLIT ? CI16_RS : CV_RS, SLAB = SPLIT ? CI16_SLAB : CV_ROWS * CV_RS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    // SXW == 5 (one x-part): a 1-D grid of 8 B workgroups, re-mapped so that the eight workgroups of a scene (4 iy-groups x
    // 2 channel halves, which share the scene's sub-volumes) run on ONE XCD and hit in its L2: workgroup i runs on XCD i % 8.
    int xp = blockIdx.x, wy = blockIdx.y, b = blockIdx.z;
    if constexpr (XW == RES) {
        const int i = blockIdx.x, nfull = (B >> 3) << 6;           // workgroups of the scenes that fill whole groups of 8
        xp = 0;
        if (i < nfull) { const int xcd = i & 7, slot = i >> 3; b = (slot >> 3) * 8 + xcd; wy = slot & 7; }
        else           { b = i >> 3; wy = i & 7; }
    }
    const int grp = wy >> 1, chh = wy & 1;
    const int x0 = xp * XW;
    const float* vol = tsdf + (size_t)b * RES * RES * RES;
    CI_T(0);

    // ---- stage the haloed sub-volume: rows (xl, yl) of 40 iz values + the two iz halo cells; outside = 0 ----
    // The 12 rows of one ix slab are 120 consecutive float4 in memory.  A thread keeps ONE (row, float4) position of a slab
    // for the whole kernel and walks the slabs (NT / 120 per pass), so an item costs two adds, not four divisions; all global
    // loads are in flight before the first LDS write, and the setup arithmetic below runs under their latency.
    constexpr int SPP = NT / 120, NPASS = (XW + 2 + SPP - 1) / SPP;          // slabs per pass, passes
    const int st_slab = tid / 120, st_e = tid - 120 * st_slab, st_yl = st_e / 10, st_q = st_e - 10 * st_yl;
    const int st_iy = 10 * grp - 1 + st_yl;
    const bool st_on = st_slab < SPP && st_iy >= 0 && st_iy < RES;
    const float* st_src = vol + ((size_t)(x0 - 1 + st_slab) * RES + st_iy) * RES + 4 * st_q;
    float4 vals[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int xl = st_slab + SPP * p, ix = x0 - 1 + xl;
        vals[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st_on && xl < XW + 2 && ix >= 0 && ix < RES)
            vals[p] = *reinterpret_cast<const float4*>(st_src + (size_t)SPP * p * RES * RES);
    }
    CI_T(50);
    float wreg[7];
    half8 wsh = {0, 0, 0, 0, 0, 0, 0, 0}, wsl = wsh;
    if constexpr (SPLIT) {
        const half8* wsp = reinterpret_cast<const half8*>(wpk);
        wsh = wsp[(2 * chh) * 64 + lane];
        wsl = wsp[(2 * chh + 1) * 64 + lane];
    } else {
#pragma unroll
        for (int s = 0; s < 7; ++s) wreg[s] = wpk[(chh * 7 + s) * 64 + lane];
    }
    const int ch = 16 * chh + j;
    const float bn = bias[ch];
    // A operand: row i = lane&15 is voxel (iy_l = i>>3, iz_l = 4*((i>>2)&1) + (i&3)); k-slot g supplies tap 4s+g
    // (tap = dx*9 + dy*3 + dz; tap 27 has zero weight and reads tap 26's voxel)
    int abase[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        int t = 4 * s + g;
        t = t > 26 ? 26 : t;
        abase[s] = (t / 9) * SLAB + ((j >> 3) + (t / 3) % 3) * RS + CV_OFF + 4 * ((j >> 2) & 1) + (j & 3) + t % 3;
    }
    // SPLIT: k-slot g of the single K = 32 step reads the voxels of taps ci16_read(g, 0..7) (giga_layout.h)
    int sbase[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t = g == 0 ? ci16_read(0, e) : g == 1 ? ci16_read(1, e) : g == 2 ? ci16_read(2, e) : ci16_read(3, e);
        sbase[e] = (t / 9) * SLAB + ((j >> 3) + (t / 3) % 3) * RS + CV_OFF + 4 * ((j >> 2) & 1) + (j & 3) + t % 3;
    }
    f32x4v acc_yz[5][5];
#pragma unroll
    for (int ip = 0; ip < 5; ++ip)
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) acc_yz[ip][zg] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const float inv = 1.0f / RES;
    const size_t img_stride = (size_t)RES * RES * CD;
    TOut* plane_xy = planes + ((size_t)1 * B + b) * img_stride;
    TOut* plane_yz = planes + ((size_t)2 * B + b) * img_stride;
    float* xzp = xz_partial + ((size_t)grp * B + b) * img_stride;
    const f32x4v bias4 = {bn, bn, bn, bn};            // the bias rides in the C operand of the first MFMA
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned xz_lane_b = ((4 * (g & 1) + 2 * (g >> 1)) * RES * CD + ch) * 4u;   // lane part of the xz-partial address (bytes): rows r = 2 (g >> 1) + {0, 1}
    // plane xz [iz][ix][c], this iy-group's share of one (ix, iz-group): the other iy row of each tile lives in lane^32.
    // Two v_permlane32_swap (plain VALU: no LDS round trip, no wait) leave rows r = 0,1 summed in lanes 0..31 and rows
    // r = 2,3 in lanes 32..63, so every lane stores two values.  `dst` = uniform part of the address.
    auto xz_store = [&](const f32x4v& part, float* dst) {
        float a02 = part[0], b02 = part[2], a13 = part[1], b13 = part[3];
        lane32_swap(a02, b02);
        lane32_swap(a13, b13);
        store_f32_saddr(dst, xz_lane_b, a02 + b02);                 // uniform base + 32-bit lane offset: no VALU address arithmetic
        store_f32_saddr(dst + RES * CD, xz_lane_b, a13 + b13);
    };
    // plane xy [iy][ix][c] of one ix: the other half of the 8 iz of each tile lives in lane^16.  v_permlane16_swap pairs two
    // iy-pairs: even 16-lane rows end up with the total of pair `ia`, odd rows with `ib`.
    auto xy_store = [&](const f32x2v (&sz)[5], TOut* dst) {
#pragma unroll
        for (int ia = 0; ia < 5; ia += 2) {
            const int ib = ia + 1 < 5 ? ia + 1 : ia;
            float sa = sz[ia][0] + sz[ia][1], sb = sz[ib][0] + sz[ib][1];
            lane16_swap(sa, sb);
            if (ia != ib || (g & 1) == 0) {
                const int iy = grp * 10 + 2 * (ia + (g & 1) * (ib - ia)) + (g >> 1);
                dst[iy * RES * CD + ch] = (TOut)((sa + sb) * inv);
            }
        }
    };
    // ---- the staged values go to LDS: row = 44 floats, iz = -1 at float 3 (so the 40 values start 16-byte aligned: one
    // ds_write_b128 per item), iz = 40 at float 44 = the unused float 0 of the next row ----
    if (st_slab < SPP) {
        float* st_dst = lds + st_slab * SLAB + st_yl * RS + CV_OFF + 1 + 4 * st_q;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (st_slab + SPP * p >= XW + 2) break;
            float* dst = st_dst + SPP * p * SLAB;
            float4 val = vals[p];
            if constexpr (SPLIT) {                     // word = hi | lo << 16
                float* v4 = reinterpret_cast<float*>(&val);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const half_t h = (half_t)v4[e];
                    const half_t l = (half_t)__builtin_fmaf((float)h, -1.0f, v4[e]);
                    const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
                    v4[e] = __builtin_bit_cast(float, w);
                }
            }
            *reinterpret_cast<float4*>(dst) = val;
            if (st_q == 0) dst[-1] = 0.f;
            if (st_q == 9) dst[4] = 0.f;
        }
    }
    CI_T(1);
    __syncthreads();
    CI_T(2);

    if constexpr (SPLIT) {
    for (int sx = 0; sx < SXW; ++sx) {
        const int ixl = wave * SXW + sx, ix = x0 + ixl;
        unsigned mw[4] = {0u, 0u, 0u, 0u};
        f32x2v sum_z[5];                  // per iy-pair: sum over the 5 iz-groups and the 4 in-lane iz (two partials)
#pragma unroll
        for (int ip = 0; ip < 5; ++ip) sum_z[ip] = f32x2v{0.f, 0.f};
        const unsigned* ldw = reinterpret_cast<const unsigned*>(lds);
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) {
            f32x4v d[5];
            CI_T(3 + 6 * sx + zg);
            // two batches (3 + 2 units) keep the operand registers low: the 100 yz accumulators stay resident
#pragma unroll
            for (int b0 = 0; b0 < 5; b0 += 3) {
                constexpr int NBATCH = 3;
                half8 ah[NBATCH], al[NBATCH];
#pragma unroll
                for (int u = 0; u < NBATCH; ++u) {
                    const int ip = b0 + u;
                    if (ip < 5) {
                        unsigned w8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) w8[e] = ldw[sbase[e] + ixl * SLAB + 2 * ip * RS + 8 * zg];
                        unsigned hw[4], lw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {   // bytes [w0.b0 w0.b1 w1.b0 w1.b1] = two hi halfs, [w0.b2 w0.b3 w1.b2 w1.b3] = two lo halfs
                            hw[q] = __builtin_amdgcn_perm(w8[2 * q + 1], w8[2 * q], 0x05040100u);
                            if constexpr (LO) lw[q] = __builtin_amdgcn_perm(w8[2 * q + 1], w8[2 * q], 0x07060302u);
                        }
                        ah[u] = __builtin_bit_cast(half8, uint4{hw[0], hw[1], hw[2], hw[3]});
                        al[u] = __builtin_bit_cast(half8, uint4{lw[0], lw[1], lw[2], lw[3]});
                    }
                }
#pragma unroll
                for (int u = 0; u < NBATCH; ++u) if (b0 + u < 5) d[b0 + u] = LO ? mfma16_16(ah[u], wsl, bias4) : bias4;
#pragma unroll
                for (int u = 0; u < NBATCH; ++u) if (LO && b0 + u < 5) d[b0 + u] = mfma16_16(al[u], wsh, d[b0 + u]);
#pragma unroll
                for (int u = 0; u < NBATCH; ++u) if (b0 + u < 5) d[b0 + u] = mfma16_16(ah[u], wsh, d[b0 + u]);
            }
            f32x4v part_y = zero4;            // per r: sum over the 5 iy-pairs of this group
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) {
                if constexpr (MASK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = (zg * 5 + ip) * 4 + r;
                        const float dv = d[ip][r];       // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0 of the vector)
                        mw[k >> 5] = __builtin_amdgcn_alignbit(mw[k >> 5], __builtin_bit_cast(unsigned, dv), 31);
                    }
                }
                const f32x4v v = {relu(d[ip][0]), relu(d[ip][1]), relu(d[ip][2]), relu(d[ip][3])};
                acc_yz[ip][zg] += v;
                // pin the accumulation here: the IR sinking pass otherwise moves it to the loop latch and keeps all 100
                // ReLU outputs of a slice live next to the 100 accumulators (spills)
                asm volatile("" : "+v"(acc_yz[ip][zg]));
                sum_z[ip] += f32x2v{v[0], v[1]} + f32x2v{v[2], v[3]};
                part_y += v;
            }
            xz_store(part_y, xzp + (8 * zg * RES + ix) * CD);
            __builtin_amdgcn_sched_barrier(0);
        }
        xy_store(sum_z, plane_xy + ix * CD);
        if constexpr (MASK) relu_mask[(((size_t)b * 8 + (2 * grp + chh)) * RES + ix) * 64 + lane] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
    }
    } else {
    // ---- fp32-input MFMA path.  On gfx950 v_mfma_f32_*_f32 and ordinary VALU work share one pipe (no co-execution, not even
    // across waves: tools/mfma_valu_overlap.hip) and every MFMA <-> VALU switch costs ~10 clocks on top (a VALU instruction
    // between two MFMAs costs 14 clocks, in a run 4.4: tools/mfma_issue_cost.hip).  So a stage (= one iz-group) is ONE
    // uninterrupted burst of 35 MFMAs (7 k-steps x 5 iy-pairs) carrying only the LDS reads of the A operands three k-steps
    // ahead (rolling window, runs on into the next stage / slice: nothing is ever waited for), followed by ONE dense run of
    // VALU work: ReLU, the three axis sums, the lane swaps and the plane stores.
    constexpr int SLICE = SLAB;
    // two LDS base addresses per k-step (iy-pairs 0..2 / 3..4), kept in registers for the whole kernel and bumped once per
    // slice: every read of the burst is base + immediate (the asm pins stop the compiler from re-deriving them inside it)
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
    int a_lo[7], a_hi[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        a_lo[s] = (int)lds_base + 4 * (abase[s] + wave * SXW * SLICE);   // absolute LDS byte addresses
        a_hi[s] = a_lo[s] + 4 * 6 * RS;
        asm volatile("" : "+v"(a_lo[s]), "+v"(a_hi[s]));
    }
    float A[7][5];
    auto load_a = [&](int s, int off) {
#pragma unroll
        for (int ip = 0; ip < 5; ++ip)
            A[s][ip] = *(lds_cfloat*)((size_t)(unsigned)((ip < 3 ? a_lo[s] : a_hi[s]) + 4 * (2 * (ip < 3 ? ip : ip - 3) * RS + off)));
    };
    load_a(0, 0); load_a(1, 0); load_a(2, 0);
    for (int sx = 0; sx < SXW; ++sx) {
        const int ix = x0 + wave * SXW + sx;
        unsigned mw[4] = {0u, 0u, 0u, 0u};
        f32x2v sum_z[5];                  // per iy-pair: sum over the 5 iz-groups and the 4 in-lane iz (two partials)
#pragma unroll
        for (int zg = 0; zg < 5; ++zg) {
            CI_T(3 + 6 * sx + zg);
            f32x4v d[5];
        #pragma unroll
            for (int s = 0; s < 7; ++s) {
                if (s + 3 < 7) load_a(s + 3, 8 * zg);
                else           load_a(s - 4, zg < 4 ? 8 * (zg + 1) : 0);           // next iz-group / next slice (bases 0..2 already moved on)
#pragma unroll
                for (int ip = 0; ip < 5; ++ip) d[ip] = mfma32_16(A[s][ip], wreg[s], s == 0 ? bias4 : d[ip]);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4v part_y;                // per r: sum over the 5 iy-pairs of this group
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) {
                if constexpr (MASK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = (zg * 5 + ip) * 4 + r;
                        const float dv = d[ip][r];       // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0 of the vector)
                        mw[k >> 5] = __builtin_amdgcn_alignbit(mw[k >> 5], __builtin_bit_cast(unsigned, dv), 31);
                    }
                }
                const f32x4v v = {relu(d[ip][0]), relu(d[ip][1]), relu(d[ip][2]), relu(d[ip][3])};
                acc_yz[ip][zg] += v;
                asm volatile("" : "+v"(acc_yz[ip][zg]));         // keep the accumulation here (see the split path)
                const f32x2v h = f32x2v{v[0], v[1]} + f32x2v{v[2], v[3]};
                sum_z[ip] = zg == 0 ? h : sum_z[ip] + h;
                part_y = ip == 0 ? v : part_y + v;
            }
            xz_store(part_y, xzp + (8 * zg * RES + ix) * CD);
            if (zg == 3) {                // k-steps 0..2 of this slice have all been read: their bases move to the next slice
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    a_lo[s] += 4 * SLICE; a_hi[s] += 4 * SLICE;
                    asm volatile("" : "+v"(a_lo[s]), "+v"(a_hi[s]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        xy_store(sum_z, plane_xy + ix * CD);
        if constexpr (MASK) relu_mask[(((size_t)b * 8 + (2 * grp + chh)) * RES + ix) * 64 + lane] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
#pragma unroll
        for (int s = 3; s < 7; ++s) {
            a_lo[s] += 4 * SLICE; a_hi[s] += 4 * SLICE;
            asm volatile("" : "+v"(a_lo[s]), "+v"(a_hi[s]));
        }
    }
    }
    // ---- plane yz: fixed-order sum of the NW waves (each holds the sum over its own slices), UR units per round ----
    f32x4v* slot = reinterpret_cast<f32x4v*>(lds);                     // [wave][unit of the round][lane]
    CI_T(40);
    float* yzp = yz_partial + ((size_t)xp * B + b) * img_stride;
    constexpr int UR = ci_red_units(NW), NRD = (25 + UR - 1) / UR;
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
        const int u0 = rd * UR, un = 25 - u0 < UR ? 25 - u0 : UR;
        __syncthreads();                                               // (round 0: every wave is done with the sub-volume)
#pragma unroll
        for (int ul = 0; ul < UR; ++ul)
            if (ul < un) slot[(wave * UR + ul) * 64 + lane] = acc_yz[(u0 + ul) / 5][(u0 + ul) % 5];
        __syncthreads();
        for (int e = tid; e < un * 64; e += NT) {
            const int ul = e >> 6, ln = e & 63;
            f32x4v sum = slot[ul * 64 + ln];
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += slot[(w * UR + ul) * 64 + ln];
            const int u = u0 + ul, ip = u / 5, zg = u % 5, lg = ln >> 4, lj = ln & 15;
            const int iy = grp * 10 + 2 * ip + (lg >> 1), iz0 = 8 * zg + 4 * (lg & 1);
            const int di = (iz0 * RES + iy) * CD + 16 * chh + lj;
            if constexpr (XW == RES) {                         // one x-part: this IS the plane (mean over all 40 ix)
#pragma unroll
                for (int r = 0; r < 4; ++r) plane_yz[di + r * RES * CD] = (TOut)(sum[r] * inv);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) yzp[di + r * RES * CD] = sum[r];
            }
        }
        CI_T(41 + rd);
    }
    CI_T(63);
}

// planes xz = (sum of the 4 iy-group partials) / 40, yz = (sum of the NXP x-part partials) / 40
template <typename TOut>
__global__ void plane_finalize_kernel(const float* __restrict__ xz_partial, const float* __restrict__ yz_partial,
                                      TOut* __restrict__ planes, int B, int nxp, unsigned* __restrict__ sync_words, int nsync) {
    const size_t per4 = (size_t)B * RES * RES * CD / 4;     // float4 elements of one plane over the batch
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // this launch always precedes the U-Net: it clears the barrier words of the persistent U-Net kernel (no separate memset)
    if (blockIdx.x == 0 && sync_words) for (int i = threadIdx.x; i < nsync; i += blockDim.x) sync_words[i] = 0u;
    if (t >= per4) return;
    const f32x4v* xs = reinterpret_cast<const f32x4v*>(xz_partial) + t;
    const f32x4v* ys = reinterpret_cast<const f32x4v*>(yz_partial) + t;
    typedef TOut out4 __attribute__((ext_vector_type(4)));      // one 8- or 16-byte store per thread and plane
    f32x4v sx = (xs[0] + xs[per4]) + (xs[2 * per4] + xs[3 * per4]);
    out4 ox;
#pragma unroll
    for (int r = 0; r < 4; ++r) ox[r] = (TOut)(sx[r] * (1.0f / RES));
    *reinterpret_cast<out4*>(planes + 4 * t) = ox;
    if (nxp > 1) {                                          // (one x-part: conv_in wrote the yz plane itself)
        f32x4v sy = ys[0];
        for (int k = 1; k < nxp; ++k) sy += ys[(size_t)k * per4];
        out4 oy;
#pragma unroll
        for (int r = 0; r < 4; ++r) oy[r] = (TOut)(sy[r] * (1.0f / RES));
        *reinterpret_cast<out4*>(planes + 2 * 4 * per4 + 4 * t) = oy;
    }
}

// ----------------------------------------------------------------------------------------------------
// Encoder driver.  Workspace carve (all NHWC, element type T):
//   P0 planes_in [3B,40,40,32] | A0 | S0 (skip 0) | Q0 [3B,20,20,32] | A1 [..,20,20,64] | S1 | Q1 [..,10,10,64]
//   | A2 [..,10,10,128] | S2 | U0 [..,20,20,64] | A3 | A4 | U1 [..,40,40,32] | A5 | A6 | YZ partials fp32
// ----------------------------------------------------------------------------------------------------
// x-parts of conv_in+project: 1 (8*B workgroups, five slices per wave) from 32 scenes up, else 5 (40*B workgroups)
int enc_nxp(int B) { return B >= 32 ? 1 : 5; }

EncWs enc_workspace(int B, int precision) {
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
    w.YZ = take((size_t)(4 + enc_nxp(B)) * B * 1600 * 32, 4);   // 4 xz partials (iy-groups) + NXP yz partials (x-parts)
    w.XZ = w.YZ;
    w.SYNC = take(MEGA_SYNC_WORDS * 4, 4); // barrier counters of the persistent U-Net kernel (8 per-XCD + 32 per-group, 128 B apart)
    w.MASK = take((size_t)B * 8 * RES * 64 * 4, 4);   // conv_in ReLU mask (training forward with GIGA_CONVIN_MASK; 10 MB at 32 scenes)
    w.total = at;
    return w;
}


// ====================================================================================================
// Persistent U-Net kernel: all 12 (13) layers of UNet.forward (encoder/unet.py:225-239) in ONE launch.
//   At 32 scenes a U-Net layer is 10-55 us of which ~7 us is per-launch floor (dispatch gap, the weight fill into LDS, the
//   first patch round trip, the tail) -- 13 launches, i.e. ~90 us of a 200-420 us encoder that no per-layer tuning removes.
//   EXPERIMENT, NOT THE PRODUCT PATH: measured 573 vs 423 us (fp32, 32 scenes) -- see the profile note.  The idea:
//   one workgroup per CU stays resident and walks the layers; between two layers there is a device-wide barrier (an
//   agent-scope release -> atomic counter -> acquire, so that the other XCDs' L2 see the layer's output), and the NEXT
//   layer's weight group is already streaming into LDS (LDS-DMA) while the workgroup waits at that barrier.
//   The layer bodies are the conv16 stages (giga_conv16.h), unchanged: same arithmetic, same results as per-layer launches.
//   Co-residency of the 256 workgroups is what the barrier needs; the spin is bounded (no hang if another kernel holds CUs:
//   the flag word behind the counter is set and the host falls back to per-layer launches).
// ====================================================================================================
// X(layer, KIND, C0, C1, COUT, H, W, NB, POOL, NB16): NB = 16-channel output blocks per unit for 4-byte operands (fp32, split:
// the resident weights of NB blocks must fit LDS next to the patches); NB16 = the same for native f16 activations, whose
// weights are half the size -- more blocks per unit = more MFMAs per A-operand read (the f16 layers are LDS-read-bound)
#define GIGA_UNET_LAYERS(X)                              \
    X(0, CONV3, 32, 0, 32, 40, 40, 2, false, 2)          \
    X(1, CONV3, 32, 0, 32, 40, 40, 2, true, 2)           \
    X(2, CONV3, 32, 0, 64, 20, 20, 1, false, 4)          \
    X(3, CONV3, 64, 0, 64, 20, 20, 1, true, 4)           \
    X(4, CONV3, 64, 0, 128, 10, 10, 1, false, 4)         \
    X(5, CONV3, 128, 0, 128, 10, 10, 1, false, 2)        \
    X(6, UPCONV, 128, 0, 64, 10, 10, 2, false, 2)        \
    X(7, CONV3, 64, 64, 64, 20, 20, 1, false, 2)         \
    X(8, CONV3, 64, 0, 64, 20, 20, 1, false, 4)          \
    X(9, UPCONV, 64, 0, 32, 20, 20, 2, false, 2)         \
    X(10, CONV3, 32, 32, 32, 40, 40, 2, false, 2)        \
    X(11, CONV3, 32, 0, 32, 40, 40, 2, false, 2)         \
    X(12, CONV1, 32, 0, 32, 40, 40, 2, false, 2)

// ----------------------------------------------------------------------------------------------------
// The U-Net as ONE persistent launch, synchronised per GROUP of 8 workgroups inside one XCD.
//   Images are independent.  The launch is 8 * S * 8 workgroups (S = 1..4 group slots per XCD; 256 = one per CU from 25 images
//   up).  The eight workgroups of a group run on ONE XCD (they find each other by ticket, see the kernel) and take a contiguous,
//   balanced share of the 3B images (one image or none up to 32 images) through all layers; a layer boundary is a barrier among
//   those 8 workgroups only, and everything they exchange goes through their XCD's own L2: the producer waits for its stores
//   (vmcnt 0: acknowledged by the L2), arrives with a workgroup-scope atomic add (performed in the L2) and spins on an L2 load;
//   a consumer reads buffers nobody on its CU has read before in this launch, so its L1 holds no older copy.  No agent-scope
//   release / acquire (write-back + invalidate of the whole L2, ~7 us) and no device-scope atomics (resolved outside the XCD,
//   ~8 us for 256 arrivals); tools/xcd_barrier.hip is the stand-alone experiment (zero stale reads).
//   8 workgroups per group because no layer has more than 8 weight groups (conv_wg_map hands them out inside the group).
//   The next layer's weights stream into LDS (LDS-DMA) while the barrier is waited for.
//   History: round 2a synchronised device-wide (lost to per-layer launches, profiles/r02e_persistent_unet_experiment.txt),
//   round 2b per XCD (32 workgroups per barrier, ~5 us each: -10 % in the f16-class modes at 8-32 scenes, an opt-in with a
//   one-stream contract because 256 blockIdx-placed workgroups had to be co-resident); round 3 per group: a barrier among 8
//   workgroups releases ~1 us (2.4 k clocks) after its last arrival, the groups drift apart instead of draining and refilling
//   the device at every layer, and ticket placement removed the co-residency contract -- the form became the DEFAULT: whole
//   encoder at one scene 94 -> 75 us (f16); at 32 scenes f16 150 -> 130, f16x3 230 -> 216, fp32 378 -> 361
//   (profiles/r03/unet_grouped_*.txt).  fp32 layers gain nothing below ~8 scenes and keep their per-layer launches there.
// ----------------------------------------------------------------------------------------------------
struct MegaArgs {
    ConvArgs layer[NCONV];
    unsigned* sync;            // words zeroed by plane_finalize_kernel: [x * 32] ticket counter of XCD x, [(8 + q) * 32] arrival counter of group q
    int nlayers;               // 12 (conv_final folded into the decoder) or 13
    unsigned wino_mask;        // bit l: layer l (a 3x3 layer of the exact-fp32 path) runs as Winograd F(2x2, 3x3) (giga_wino.h)
};
constexpr int MEGA_GROUP = 8;                     // workgroups per group
constexpr int MEGA_SLOTS = 4;                     // group slots per XCD at most: 8 x 4 x 8 = 256 workgroups, one per CU
constexpr int MEGA_GROUP_MAX_IMG = 32;            // default form of the f16-class modes up to 32 images (one image per group)
constexpr int MEGA_NW = CONV_NW;                  // waves per workgroup of the persistent kernel
// A barrier wait is bounded by WALL CLOCK (s_memrealtime, 100 MHz): 20 s.  A group's partners can be late for honest reasons -- a long
// kernel of another stream holding the CUs they need -- and a spin COUNT (rounds 2-3: 2^24 polls, a few seconds) does not
// distinguish that from a group that can never fill.
constexpr unsigned long long MEGA_SPIN_TICKS = 20ull * 100000000ull;
constexpr int MEGA_MAX_IN_FLIGHT = 4;             // persistent launches in flight per device (see persistent_slot below)

template <typename T, int MATH, bool WINO = false>
constexpr size_t mega_lds_bytes() {
    size_t m = 0;
#define X(l, KIND, C0, C1, COUT, H, W, NB, POOL, NB16) \
    { constexpr size_t v = conv_lds_bytes<T, KIND, C0, C1, COUT, H, W, (sizeof(T) == 2 ? NB16 : NB), MATH, (WINO ? WINO_NW : CONV_NW)>(); m = v > m ? v : m; \
      if constexpr (WINO && KIND == CONV3) { constexpr size_t u = wino_lds_bytes<C0, C1, H, W>(); m = u > m ? u : m; } }
    GIGA_UNET_LAYERS(X)
#undef X
    return m;
}

#ifdef GIGA_TRACE
static __device__ long long g_mega_trace[8][32];          // [workgroup 0..7][2 * layer: arrival, release]
#endif
__device__ __forceinline__ void xcd_barrier(unsigned* counter, unsigned target, int idx, int trace_row) {
#ifdef GIGA_TRACE
    if (threadIdx.x == 0 && trace_row >= 0 && trace_row < 8 && idx < 16) g_mega_trace[trace_row][2 * idx] = __builtin_amdgcn_s_memtime();
#endif
    // (ONE poller per workgroup.  Letting every wave poll for itself -- so that each requests its next patch the moment it sees the
    //  release -- measured no gain for groups of 8 workgroups and made a barrier among 32 workgroups slower: 384 pollers on one L2 line.)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // performed in this XCD's L2
        unsigned spins = 0;
        unsigned long long t0 = 0;                             // (the clock is read every 1024 polls only: the read itself is slow)
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // an L2 read
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz, independent of the shader clock
                if (t0 == 0) t0 = now;
                else if (now - t0 > MEGA_SPIN_TICKS) __builtin_trap();               // fail loudly, never hang the device
            }
        }
    }
    __syncthreads();
#ifdef GIGA_TRACE
    if (threadIdx.x == 0 && trace_row >= 0 && trace_row < 8 && idx < 16) g_mega_trace[trace_row][2 * idx + 1] = __builtin_amdgcn_s_memtime();
#endif
}

// the same barrier in two halves: arrive (after the workgroup's stores are acknowledged), then -- after the caller has put its next
// weight fill in flight -- wait.  What sits between the two does not delay the other members of the group.
__device__ __forceinline__ void xcd_arrive(unsigned* counter) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // performed in this XCD's L2
}
__device__ __forceinline__ void xcd_wait(unsigned* counter, unsigned target) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // an L2 read
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > MEGA_SPIN_TICKS) __builtin_trap();               // fail loudly, never hang the device
            }
        }
    }
    __syncthreads();
}

// the image range [img0, img0 + n) of one layer's operands
template <typename T, int KIND, int C0, int C1, int COUT, int H, int W>
__device__ __forceinline__ ConvArgs conv_image_range(ConvArgs a, int img0, int n) {
    constexpr int IH = KIND == DOWN ? 2 * H : H, IW = KIND == DOWN ? 2 * W : W;
    constexpr int OH = KIND == UPCONV ? 2 * H : H, OW = KIND == UPCONV ? 2 * W : W;
    const size_t s0 = (size_t)IH * IW * (a.cs0 ? a.cs0 : C0), s1 = (size_t)IH * IW * (a.cs1 ? a.cs1 : C1);
    a.in0 = reinterpret_cast<const T*>(a.in0) + img0 * s0;
    if (a.in1) a.in1 = reinterpret_cast<const T*>(a.in1) + img0 * s1;
    a.out = reinterpret_cast<T*>(a.out) + (size_t)img0 * OH * OW * COUT;
    if (a.out_pool) a.out_pool = reinterpret_cast<T*>(a.out_pool) + (size_t)img0 * (H / 2) * (W / 2) * COUT;
    if (a.out_nchw) a.out_nchw += (size_t)img0 * COUT * H * W;
    if (a.mask) a.mask += (size_t)img0 * OH * OW * COUT;
    a.nimg = n;
    if (img0 != 0) a.trace_id = -2;                  // (diagnostic builds trace the group that holds image 0)
    return a;
}

// WINO (exact fp32 only): the instantiation whose 3x3 layers may run as Winograd stages (giga_wino.h).  It launches WINO_NW = 8 waves
// per workgroup -- those stages hold 64 accumulators + a transformed half-chunk + a prefetched chunk per lane, ~230 registers, which
// three waves per SIMD (168) cannot -- and its direct stages (ConvTranspose, 1x1, unselected 3x3 layers) walk their units with 8 waves.
template <typename T, int MATH, bool WINO = false>
__global__ __launch_bounds__((WINO ? WINO_NW : MEGA_NW) * 64) void unet_mega_kernel(MegaArgs m) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // A workgroup finds its place by TICKET, not by blockIdx: it reads the XCD it runs on from the hardware (XCC_ID) and draws a
    // number from that XCD's ticket counter; ticket t is member t % 8 of group slot t / 8 on that XCD.
    //   * The barriers are XCD-local (workgroup-scope atomics, no agent-scope release / acquire), i.e. only correct if the
    //     workgroups that share a counter share an L2 -- true by construction here, whatever order the dispatcher uses.  (Rounds
    //     2-3a assumed "workgroup i runs on XCD i % 8" and checked it; the dispatcher's round-robin pointer carries over from the
    //     previous launch, so even that map is rotated after a grid that is not a multiple of 8.)  What is still assumed is that
    //     every XCD receives gridDim / 8 workgroups: a group that never fills traps after a few seconds instead of returning stale data.
    //   * The members of a group are the first eight workgroups of the launch that became RESIDENT on their XCD, so a launch has
    //     at most one unfilled group per XCD, and filled groups depend on nobody: several of these launches in flight at once
    //     (other streams, other processes) cannot hold each other's CUs in a cycle -- which is what made the blockIdx-based form
    //     an opt-in with a one-stream contract.
    __shared__ unsigned s_place[2];
    if (threadIdx.x == 0) {
        const unsigned xcc = (unsigned)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7);     // HW_REG_XCC_ID[3:0]
        s_place[0] = xcc;
        s_place[1] = __hip_atomic_fetch_add(m.sync + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // performed in this XCD's L2
    }
    __syncthreads();
    const int xcd = (int)s_place[0], ticket = (int)s_place[1];
    const int nq = (int)gridDim.x >> 3, slot = ticket / MEGA_GROUP, q = slot * 8 + xcd, nimg = m.layer[0].nimg;   // nq = 8 * slots = #groups
    if (slot >= nq >> 3) return;                                            // (an XCD that was handed more than its share)
    const int img0 = (q * nimg + nq - 1) / nq, per = ((q + 1) * nimg + nq - 1) / nq - img0;   // group q of nq: a contiguous, balanced share of the images (group 0 is never empty)
    if (per == 0) return;                                                   // (before any barrier)
    const int block = ticket - slot * MEGA_GROUP, nblocks = MEGA_GROUP;     // this workgroup's place inside its group
    unsigned* counter = m.sync + (8 + q) * 32;
    unsigned epoch = 0;
    // a 3x3 layer whose bit is set in m.wino_mask runs as Winograd F(2x2, 3x3) (WINO instantiation only: exact fp32), in CIN / 64 K-passes
    auto is_wino = [&](auto kind_tag, int l) { return WINO && decltype(kind_tag)::value == CONV3 && (m.wino_mask >> l & 1u); };
#define X(l, KIND, C0, C1, COUT, H, W, NB, POOL, NB16)                                                                     \
    if (l < m.nlayers) {                                                                                                   \
        const ConvArgs a = conv_image_range<T, KIND, C0, C1, COUT, H, W>(m.layer[l], img0, per);                           \
        bool direct = true;                                                                                                \
        if constexpr (WINO && KIND == CONV3) {                                                                             \
            if (is_wino(std::integral_constant<int, KIND>{}, l)) {                                                         \
                direct = false;                                                                                            \
                _Pragma("unroll") for (int k = 0; k < wino_kpass(C0 + C1); ++k) {                                          \
                    if (l == 0 || k > 0) wino_fill<C0, C1, COUT>(a, smem, block, nblocks, k);                              \
                    wino_run<C0, C1, COUT, H, W, POOL, true>(a, smem, block, nblocks, k);                                  \
                    __builtin_amdgcn_s_waitcnt(0x0F70);                                                                    \
                    __syncthreads();                                                                                       \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
        if constexpr (WINO && KIND == UPCONV) {                                                                            \
            if (m.wino_mask >> l & 1u) {                                                                                   \
                direct = false;                                                                                            \
                up_run<C0, COUT, H, W>(a, smem, block, nblocks);                                                           \
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                                        \
                __syncthreads();                                                                                           \
            }                                                                                                              \
        }                                                                                                                  \
        if (direct) {                                                                                                      \
        if (l == 0) conv16_fill<T, KIND, C0, C1, COUT, (sizeof(T) == 2 ? NB16 : NB), MATH>(a, smem, block, nblocks);                \
        conv16_run<T, KIND, C0, C1, COUT, H, W, (sizeof(T) == 2 ? NB16 : NB), POOL, KIND == CONV3, MATH, (WINO ? WINO_NW : CONV_NW)>(a, smem, block, nblocks); \
        __builtin_amdgcn_s_waitcnt(0x0F70);            /* vmcnt(0): this wave's output stores are in the L2 */              \
        __syncthreads();                               /* ... everyone's, and everyone has left the weights in LDS */       \
        }                                                                                                                  \
    }
    // after layer l: request layer l+1's weights (they land while the barrier is waited for), then the XCD barrier
#define NEXT(l, KIND, C0, C1, COUT, H, W, NB, POOL, NB16)                                                                  \
    if (l < m.nlayers) {                                                                                                   \
        bool direct = true;                                                                                                \
        if constexpr (WINO && KIND == CONV3) {                                                                             \
            if (is_wino(std::integral_constant<int, KIND>{}, l)) { direct = false; wino_fill<C0, C1, COUT>(m.layer[l], smem, block, nblocks, 0); } \
        }                                                                                                                  \
        if constexpr (WINO && KIND == UPCONV) {                                                                            \
            if (m.wino_mask >> l & 1u) { direct = false; up_fill<C0, COUT>(m.layer[l], smem, block, nblocks); }            \
        }                                                                                                                  \
        if (direct) conv16_fill<T, KIND, C0, C1, COUT, (sizeof(T) == 2 ? NB16 : NB), MATH>(m.layer[l], smem, block, nblocks);       \
        xcd_barrier(counter, ++epoch * (unsigned)nblocks, l, img0 == 0 ? block : -1);                                                                 \
    }
    X(0, CONV3, 32, 0, 32, 40, 40, 2, false, 2)
    NEXT(1, CONV3, 32, 0, 32, 40, 40, 2, true, 2)
    X(1, CONV3, 32, 0, 32, 40, 40, 2, true, 2)
    NEXT(2, CONV3, 32, 0, 64, 20, 20, 1, false, 4)
    X(2, CONV3, 32, 0, 64, 20, 20, 1, false, 4)
    NEXT(3, CONV3, 64, 0, 64, 20, 20, 1, true, 4)
    X(3, CONV3, 64, 0, 64, 20, 20, 1, true, 4)
    NEXT(4, CONV3, 64, 0, 128, 10, 10, 1, false, 4)
    X(4, CONV3, 64, 0, 128, 10, 10, 1, false, 4)
    NEXT(5, CONV3, 128, 0, 128, 10, 10, 1, false, 2)
    X(5, CONV3, 128, 0, 128, 10, 10, 1, false, 2)
    NEXT(6, UPCONV, 128, 0, 64, 10, 10, 2, false, 2)
    X(6, UPCONV, 128, 0, 64, 10, 10, 2, false, 2)
    NEXT(7, CONV3, 64, 64, 64, 20, 20, 1, false, 2)
    X(7, CONV3, 64, 64, 64, 20, 20, 1, false, 2)
    NEXT(8, CONV3, 64, 0, 64, 20, 20, 1, false, 4)
    X(8, CONV3, 64, 0, 64, 20, 20, 1, false, 4)
    NEXT(9, UPCONV, 64, 0, 32, 20, 20, 2, false, 2)
    X(9, UPCONV, 64, 0, 32, 20, 20, 2, false, 2)
    NEXT(10, CONV3, 32, 32, 32, 40, 40, 2, false, 2)
    X(10, CONV3, 32, 32, 32, 40, 40, 2, false, 2)
    NEXT(11, CONV3, 32, 0, 32, 40, 40, 2, false, 2)
    X(11, CONV3, 32, 0, 32, 40, 40, 2, false, 2)
    NEXT(12, CONV1, 32, 0, 32, 40, 40, 2, false, 2)
    X(12, CONV1, 32, 0, 32, 40, 40, 2, false, 2)
#undef NEXT
#undef X
}

// ----------------------------------------------------------------------------------------------------
// The DATA-GRADIENT chain of the U-Net (training backward, giga_encoder_bwd.hip) as ONE persistent launch with the forward's launch
// structure: thirteen data-gradient convolutions (conv16 with the flipped / transposed fragments of the backward blob; the ReLU mask
// of the layer below applied in the epilogue) and the two max-pool + skip-concat backward passes as stages of one kernel -- groups
// of 8 workgroups placed by ticket inside one XCD, a group walks its share of the 3B images through all fifteen stages with a
// barrier among those 8 only.  Replaces 13 + 2 launches (181 + 23 us at 32 scenes in the bf16 step).  The weight gradients, which
// need only what this chain leaves in memory, run after it.
// Stage order: L12 L11 L10 L9 L8 L7 L6 L5 L4 pool1 L3 L2 pool0 L1 L0  (reference encoder/unet.py:225-239 reversed).
// ----------------------------------------------------------------------------------------------------
#define GIGA_UNET_BWD_STAGES(X, P)                       \
    X(0, CONV1, 32, 0, 32, 40, 40, 2)                    \
    X(1, CONV3, 32, 0, 32, 40, 40, 2)                    \
    X(2, CONV3, 32, 0, 64, 40, 40, 2)                    \
    X(3, DOWN, 32, 0, 64, 20, 20, 2)                     \
    X(4, CONV3, 64, 0, 64, 20, 20, 1)                    \
    X(5, CONV3, 64, 0, 128, 20, 20, 1)                   \
    X(6, DOWN, 64, 0, 128, 10, 10, 1)                    \
    X(7, CONV3, 128, 0, 128, 10, 10, 1)                  \
    X(8, CONV3, 128, 0, 64, 10, 10, 1)                   \
    P(0)                                                 \
    X(9, CONV3, 64, 0, 64, 20, 20, 1)                    \
    X(10, CONV3, 64, 0, 32, 20, 20, 2)                   \
    P(1)                                                 \
    X(11, CONV3, 32, 0, 32, 40, 40, 2)                   \
    X(12, CONV3, 32, 0, 32, 40, 40, 2)
template <int MATH>
constexpr size_t bwd_mega_lds_bytes() {
    size_t m = 0;
#define X(k, KIND, C0, C1, COUT, H, W, NB) { constexpr size_t v = conv_lds_bytes<float, KIND, C0, C1, COUT, H, W, NB, MATH>(); m = v > m ? v : m; }
#define P(k)
    GIGA_UNET_BWD_STAGES(X, P)
#undef X
#undef P
    return m;
}
// the pool stage for the images [img0, img0 + per) of a group, dealt over the group's threads (one thread per four channels)
__device__ __forceinline__ void bwd_pool_stage(const BwdPool& p, int img0, int per, int block, int nblocks) {
    const int C4 = p.C >> 2;
    const unsigned total = (unsigned)per * p.H * p.W * C4;
    for (unsigned i = (unsigned)block * blockDim.x + threadIdx.x; i < total; i += (unsigned)nblocks * blockDim.x) {
        const int c = 4 * (int)(i % C4);
        const unsigned pl = i / C4;                                            // pixel inside the group's images
        const int x = (int)(pl % p.W), y = (int)((pl / p.W) % p.H);
        const size_t img = (size_t)img0 + pl / ((unsigned)p.W * p.H);
        const size_t pix = (size_t)img0 * p.H * p.W + pl;
        const size_t qi = ((img * (p.H / 2) + y / 2) * (p.W / 2) + x / 2) * p.C + c;
        const float4 s4 = *reinterpret_cast<const float4*>(p.S + pix * p.C + c);
        const float4 d = *reinterpret_cast<const float4*>(p.dcat + pix * p.cs + p.coff + c);
        const float4 q = *reinterpret_cast<const float4*>(p.Q + qi);
        const float4 g = *reinterpret_cast<const float4*>(p.dQ + qi);
        float4 o;
        o.x = s4.x > 0.f ? d.x + (s4.x == q.x ? g.x : 0.f) : 0.f;
        o.y = s4.y > 0.f ? d.y + (s4.y == q.y ? g.y : 0.f) : 0.f;
        o.z = s4.z > 0.f ? d.z + (s4.z == q.z ? g.z : 0.f) : 0.f;
        o.w = s4.w > 0.f ? d.w + (s4.w == q.w ? g.w : 0.f) : 0.f;
        *reinterpret_cast<float4*>(p.dS + pix * p.C + c) = o;
    }
}
template <int MATH>
__global__ __launch_bounds__(MEGA_NW * 64) void unet_dgrad_mega_kernel(BwdMegaArgs m) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ unsigned s_place[2];                    // placement by ticket: see unet_mega_kernel
    if (threadIdx.x == 0) {
        const unsigned xcc = (unsigned)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7);
        s_place[0] = xcc;
        s_place[1] = __hip_atomic_fetch_add(m.sync + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const int xcd = (int)s_place[0], ticket = (int)s_place[1];
    const int nq = (int)gridDim.x >> 3, slot = ticket / MEGA_GROUP, q = slot * 8 + xcd, nimg = m.layer[0].nimg;
    if (slot >= nq >> 3) return;
    const int img0 = (q * nimg + nq - 1) / nq, per = ((q + 1) * nimg + nq - 1) / nq - img0;
    if (per == 0) return;
    const int block = ticket - slot * MEGA_GROUP, nblocks = MEGA_GROUP;
    unsigned* counter = m.sync + (8 + q) * 32;
    unsigned epoch = 0;
    // X(k): [group barrier; stage k's weights stream into LDS while it is waited for] run the stage; the stores are acknowledged and
    // every wave has left the weights.  P(k): group barrier, the pool pass, stores acknowledged.
#define X(k, KIND, C0, C1, COUT, H, W, NB)                                                                                  \
    {                                                                                                                       \
        const ConvArgs a = conv_image_range<float, KIND, C0, C1, COUT, H, W>(m.layer[k], img0, per);                        \
        conv16_fill<float, KIND, C0, C1, COUT, NB, MATH>(a, smem, block, nblocks);                                          \
        if (k > 0) xcd_barrier(counter, ++epoch * (unsigned)nblocks, k, -1);                                                \
        conv16_run<float, KIND, C0, C1, COUT, H, W, NB, false, false, MATH>(a, smem, block, nblocks);                       \
        __builtin_amdgcn_s_waitcnt(0x0F70);                                                                                 \
        __syncthreads();                                                                                                    \
    }
#define P(k)                                                                                                                \
    {                                                                                                                       \
        xcd_barrier(counter, ++epoch * (unsigned)nblocks, 13 + k, -1);                                                      \
        bwd_pool_stage(m.pool[k], img0, per, block, nblocks);                                                               \
        __builtin_amdgcn_s_waitcnt(0x0F70);                                                                                 \
        __syncthreads();                                                                                                    \
    }
    GIGA_UNET_BWD_STAGES(X, P)
#undef X
#undef P
}

// ----------------------------------------------------------------------------------------------------
// The f16-class U-Net on conv32 (giga_conv32.h): the same persistent launch -- groups of 8 workgroups placed by ticket inside one
// XCD, a barrier among those 8 per layer -- but a member of a group owns an eighth of the ROWS of the group's stacked images
// (all output channels), keeps its sub-band of the layer's input resident in LDS and its weights in registers.  4 waves of up to
// 512 VGPRs.  The layer table (GIGA_UNET32_LAYERS) lives in giga_conv32_geom.h.
// ----------------------------------------------------------------------------------------------------
template <int MODE, bool FUSE>
__global__ __launch_bounds__(C32_NW * 64) void unet32_mega_kernel(MegaArgs m) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // placement by ticket: see unet_mega_kernel
    __shared__ unsigned s_place[2];
    if (threadIdx.x == 0) {
        const unsigned xcc = (unsigned)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7);     // HW_REG_XCC_ID[3:0]
        s_place[0] = xcc;
        s_place[1] = __hip_atomic_fetch_add(m.sync + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const int xcd = (int)s_place[0], ticket = (int)s_place[1];
    const int nq = (int)gridDim.x >> 3, slot = ticket / MEGA_GROUP, q = slot * 8 + xcd, nimg = m.layer[0].nimg;
    if (slot >= nq >> 3) return;
    int img0, per;
    c32_group_images(q, nq, nimg, img0, per);
    if (per == 0) return;                                                   // (before any barrier)
    const int block = ticket - slot * MEGA_GROUP;                           // this workgroup's place inside its group
    unsigned* counter = m.sync + (8 + q) * 32;
    unsigned epoch = 0;
    // FIRST(l): request the weights of layer l, no barrier.  NEXT(l): arrive at the group barrier, request layer l's weights
    // (LDS-DMA: they land while the barrier is waited for), then wait.  RUN(l): stage / MFMA / store; then every wave's stores
    // are acknowledged by the L2 and every wave has left the LDS image.
#define NEXT(l)                                                                                                    \
    using G##l = typename U32Layer<MODE, l>::G;                                                                    \
    if (l < m.nlayers) {                                                                                           \
        xcd_arrive(counter);                                                                                       \
        c32_fill<G##l>(m.layer[l], smem, block);                                                                   \
        MEGA32_T(l, 5);                                                                                            \
        xcd_wait(counter, ++epoch * (unsigned)MEGA_GROUP);                                                         \
        MEGA32_T(l, 6);                                                                                            \
    }
#define RUN(l)                                                                                                     \
    if (l < m.nlayers) {                                                                                           \
        c32_run<G##l, U32Layer<MODE, l>::RELU>(c32_image_range<G##l>(m.layer[l], img0, per), smem, block);         \
        __builtin_amdgcn_s_waitcnt(0x0F70);            /* vmcnt(0): this wave's output stores are in the L2 */      \
        __syncthreads();                                                                                           \
        MEGA32_T(l, 4);                                                                                            \
    }
#ifdef GIGA_TRACE
#define MEGA32_T(l, idx) do { if (img0 == 0 && block == 0 && (threadIdx.x & 63) == 0) \
        g_c32_trace[l][threadIdx.x >> 6][idx] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MEGA32_T(l, idx) do {} while (0)
#endif
    // PAIR(a, b): two same-resolution 3x3 layers without the barrier between them (C32Pair) where the mode's weights fit side by side
#define PAIR(la, lb, FIRSTA)                                                                                       \
    using G##la = typename U32Layer<MODE, la>::G;                                                                  \
    using G##lb = typename U32Layer<MODE, lb>::G;                                                                  \
    if constexpr (FUSE && c32_pair_ok<G##la, G##lb>()) {                                                           \
        if (!(FIRSTA)) xcd_arrive(counter);                                                                        \
        c32_fill_pair<G##la, G##lb>(m.layer[la], m.layer[lb], smem, block);                                        \
        if (!(FIRSTA)) { MEGA32_T(la, 5); xcd_wait(counter, ++epoch * (unsigned)MEGA_GROUP); MEGA32_T(la, 6); }    \
        c32_run_pair<G##la, G##lb, U32Layer<MODE, la>::RELU, U32Layer<MODE, lb>::RELU>(                            \
            c32_image_range<G##la>(m.layer[la], img0, per), c32_image_range<G##lb>(m.layer[lb], img0, per), smem, block); \
        __builtin_amdgcn_s_waitcnt(0x0F70);                                                                        \
        __syncthreads();                                                                                           \
        MEGA32_T(lb, 4);                                                                                           \
    } else {                                                                                                       \
        if (FIRSTA) { c32_fill<G##la>(m.layer[la], smem, block); } else { NEXT_(la) }                              \
        RUN(la) NEXT_(lb) RUN(lb)                                                                                  \
    }
#define NEXT_(l)                                                                                                   \
    {                                                                                                              \
        xcd_arrive(counter);                                                                                       \
        c32_fill<G##l>(m.layer[l], smem, block);                                                                   \
        MEGA32_T(l, 5);                                                                                            \
        xcd_wait(counter, ++epoch * (unsigned)MEGA_GROUP);                                                         \
        MEGA32_T(l, 6);                                                                                            \
    }
    PAIR(0, 1, true)
    PAIR(2, 3, false)
    NEXT(4) RUN(4)
    NEXT(5) RUN(5)
    NEXT(6) RUN(6)
    NEXT(7) RUN(7)
    NEXT(8) RUN(8)
    NEXT(9) RUN(9)
    PAIR(10, 11, false)
    NEXT(12) RUN(12)
#undef PAIR
#undef NEXT_
#undef NEXT
#undef RUN
}

// ----------------------------------------------------------------------------------------------------
// How many persistent U-Net launches may be in flight on one device.  A launch holds at most ONE unfilled group (<= 7 workgroups)
// per XCD while its remaining workgroups wait to become resident, and an XCD has 32 workgroup slots for these kernels (one per CU:
// 100-160 KiB of LDS each).  Four launches can park at most 4 x 7 = 28 < 32 slots in unfilled groups, so some group can always
// fill and finish; five or more could park 35 > 32 and every barrier would run into its time-out.  The library therefore keeps,
// per device, the completion event of the last persistent launch of every stream that has one pending; a call on a stream that
// would be the FIFTH with an unfinished persistent launch takes one launch per layer instead (same results; launches queued on one
// stream run one after the other and count once).  This is process-local: other PROCESSES sharing the device are not
// seen (include/giga_hip.h states the bound).  Streams that are being captured into a hipGraph are not tracked (an event cannot
// be queried there): a captured call keeps the persistent form, and replays on several streams at once are the caller's to bound.
// ----------------------------------------------------------------------------------------------------
struct MegaSlots {                                 // per device: the last persistent launch of up to MEGA_TRACKED streams
    static constexpr int MEGA_TRACKED = 16;
    std::atomic_flag busy = ATOMIC_FLAG_INIT;
    hipStream_t stream[MEGA_TRACKED] = {};
    hipEvent_t ev[MEGA_TRACKED] = {};
    bool used[MEGA_TRACKED] = {};                  // ev[i] records stream[i]'s last persistent launch
    bool pending[MEGA_TRACKED] = {};               // slot i is RESERVED for a launch on stream[i] that has not been recorded yet
};
static MegaSlots g_mega_slots[16];
// Launches on ONE stream run one after the other, so what counts is the number of STREAMS whose last persistent launch has not
// finished.  Returns the table slot to record after the launch (>= 0), -1 if the persistent form must not be used now (four other
// streams are busy with one, or the table is full), -2 if untracked (the stream is being captured).
static int persistent_slot(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return -2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
    MegaSlots& m = g_mega_slots[dev];
    while (m.busy.test_and_set(std::memory_order_acquire)) {}
    // The slot is RESERVED here, under the lock (pending = true, stream = s), and reservations count like launches in flight: two
    // host threads on different streams cannot both see "three others" and both take the same spare slot (the event of the second
    // would then overwrite the first's and leave a launch untracked -- exactly in the multi-stream case the bound exists for).
    int mine = -1, spare = -1, others = 0;
    for (int i = 0; i < MegaSlots::MEGA_TRACKED; ++i) {
        if ((m.used[i] || m.pending[i]) && m.stream[i] == s) { mine = i; continue; }
        if (m.used[i] && !m.pending[i] && hipEventQuery(m.ev[i]) == hipSuccess) m.used[i] = false;   // that stream's last launch has finished
        if (m.used[i] || m.pending[i]) ++others;
        else if (spare < 0) spare = i;
    }
    int slot = mine >= 0 ? mine : spare;
    if (others >= MEGA_MAX_IN_FLIGHT || slot < 0) slot = -1;
    else if (!m.ev[slot] && hipEventCreateWithFlags(&m.ev[slot], hipEventDisableTiming) != hipSuccess) slot = -1;
    if (slot >= 0) { m.pending[slot] = true; m.stream[slot] = s; }
    m.busy.clear(std::memory_order_release);
    return slot;
}
// after the launch: record its completion event and turn the reservation into a tracked launch (a failed record drops the
// reservation; a slot that tracked an earlier launch of the stream keeps tracking that one)
static void persistent_launched(int slot, hipStream_t s) {
    if (slot < 0) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
    MegaSlots& m = g_mega_slots[dev];
    while (m.busy.test_and_set(std::memory_order_acquire)) {}
    if (hipEventRecord(m.ev[slot], s) == hipSuccess) { m.used[slot] = true; m.stream[slot] = s; }
    m.pending[slot] = false;
    m.busy.clear(std::memory_order_release);
}

// after hipDeviceReset: the recorded events belong to a context that is gone -- drop them (not destroyed: their device is) and start over
void persistent_forget() {
    for (MegaSlots& m : g_mega_slots) {
        while (m.busy.test_and_set(std::memory_order_acquire)) {}
        for (int i = 0; i < MegaSlots::MEGA_TRACKED; ++i) { m.used[i] = false; m.pending[i] = false; m.ev[i] = nullptr; m.stream[i] = nullptr; }
        m.busy.clear(std::memory_order_release);
    }
}

// The data-gradient chain as one persistent launch (unet_dgrad_mega_kernel).  Returns 1 if it was launched, 0 if the persistent form
// is not to be used for this call (small batch, partitioned device, GIGA_UNET_PERSIST=0, four persistent launches already in flight
// on other streams): the caller then issues one launch per stage.  `sync`: MEGA_SYNC_WORDS words of scratch, zeroed here.
int launch_unet_dgrad_mega(BwdMegaArgs m, bool bf16, hipStream_t s) {
    static const int env_persist = [] { const char* e = getenv("GIGA_UNET_PERSIST"); return e ? atoi(e) : -1; }();
    const int nimg = m.layer[0].nimg;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (cus != 256 || env_persist == 0 || (env_persist < 0 && nimg < 24)) return 0;
    const int slot = persistent_slot(s);
    if (slot == -1) return 0;
    for (int k = 0; k < 13; ++k) m.layer[k].xcd_local = 0;                    // (the kernel hands every group its images itself)
    if (hipMemsetAsync(m.sync, 0, MEGA_SYNC_WORDS * sizeof(unsigned), s) != hipSuccess) { persistent_launched(slot, s); return -10; }
    const int slots = (nimg + 7) / 8 < MEGA_SLOTS ? (nimg + 7) / 8 : MEGA_SLOTS;
    const unsigned grid = 8u * (unsigned)slots * MEGA_GROUP;
    if (bf16) {
        constexpr size_t lds = bwd_mega_lds_bytes<MATH_BF16>();
        static_assert(lds <= 160 * 1024, "LDS budget of the persistent data-gradient kernel");
        auto kern = unet_dgrad_mega_kernel<MATH_BF16>;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
        GIGA_LAUNCH(kern, dim3(grid), dim3(MEGA_NW * 64), lds, s, m);
    } else {
        constexpr size_t lds = bwd_mega_lds_bytes<MATH_NATIVE>();
        static_assert(lds <= 160 * 1024, "LDS budget of the persistent data-gradient kernel");
        auto kern = unet_dgrad_mega_kernel<MATH_NATIVE>;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
        GIGA_LAUNCH(kern, dim3(grid), dim3(MEGA_NW * 64), lds, s, m);
    }
    persistent_launched(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : -10;
}

// probe: if probe_stage == k, ev0/ev1 (hipEvent_t) are recorded right before / after launch k
// (k = 0 conv_in+project, 1 plane_finalize, 2..14 = U-Net layers 0..12).
struct Probe { int stage; hipEvent_t ev0, ev1; };

// what the last encoder call of this process ran its U-Net on (giga_encoder_last_path: tests pin the kernel-choice flags to it)
static std::atomic<int> g_last_unet_path{0};

template <typename T, int MATH = MATH_NATIVE>
static int encoder_run(const float* tsdf, const uint8_t* blob, void* planes_nhwc, float* planes_nchw, int B,
                       uint8_t* ws, hipStream_t s, const Probe& pr, bool fold_final, int persist, int conv32, bool keep_mask = false,
                       unsigned wino_req = 0) {
    constexpr bool SPLIT = MATH == MATH_SPLIT;
    int stage_no = 0;
    auto pre = [&]() { if (pr.stage == stage_no) (void)hipEventRecord(pr.ev0, s); };
    auto post = [&]() { if (pr.stage == stage_no) (void)hipEventRecord(pr.ev1, s); ++stage_no; };
    constexpr int precision = sizeof(T) == 2 ? 1 : 0;
    const PackOff ko = pack_offsets();
    const EncWs w = enc_workspace(B, precision);
    T* P0 = reinterpret_cast<T*>(ws + w.P0);
    const size_t per = (size_t)B * RES * RES * CD;
    float* XZP = reinterpret_cast<float*>(ws + w.YZ);
    float* YZP = XZP + 4 * per;
    const int nxp = enc_nxp(B);
    // conv_in arithmetic: fp32 MFMA (precision 0), the f16 MFMA with hi only (plain f16) or with f16x3 split operands (precision 2).
    // The bf16 training forward (MATH_BF16): PLAIN f16 operands when it also stores the ReLU mask for the backward (GIGA_CONVIN_MASK:
    // the backward then uses the forward's own decisions; f16's 11 bits are finer than the bf16 rounding the next layer applies to
    // this kernel's output; -17 us per step against the split form), else the f16x3 form -- fp32-grade (<= 1e-5 of the fp32 kernel),
    // so that a backward that RECOMPUTES the mask in fp32 agrees with it.  The operand image (convin_ws) is rebuilt on the device with
    // the other derived images (giga_derive_bf16_fragments).  GIGA_BF16_CONVIN = 0: fp32 MFMA, 1: f16x3 always, 2 (default): as above.
    static const int bf_ci16 = [] { const char* e = getenv("GIGA_BF16_CONVIN"); return e ? atoi(e) : 2; }();   // 0 fp32, 1 f16x3, 2 plain f16 (default)
    uint4* mask_out = keep_mask ? reinterpret_cast<uint4*>(ws + w.MASK) : nullptr;
    auto run_convin = [&](auto f16c, auto lo, auto msk) {
        constexpr bool CI_F16 = decltype(f16c)::value, CI_LO = decltype(lo)::value, MK = decltype(msk)::value;
        const float* cw = reinterpret_cast<const float*>(blob + (CI_F16 ? ko.convin_ws : ko.convin_w));
        const float* cb = reinterpret_cast<const float*>(blob + ko.convin_b);
        // 8 waves x 5 slices.  (4 waves x 10 slices -- one wave per SIMD, half the yz partials -- measured slower for the fp32
        // path, 50.5 vs 47.3 us at 32 scenes: the second wave of a SIMD hides the slice-boundary and LDS-issue bubbles.)
        constexpr int NW = 8;
        if (nxp == 1) {
            auto kern = convin_project_kernel<T, RES / NW, CI_F16, NW, CI_LO, MK>;
            constexpr size_t lds = ci_lds_bytes(RES, NW, CI_F16);
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
            GIGA_LAUNCH(kern, dim3(8 * B), dim3(NW * 64), lds, s, tsdf, cw, cb, P0, XZP, YZP, B, mask_out);
        } else {
            auto kern = convin_project_kernel<T, 8 / NW, CI_F16, NW, CI_LO, MK>;
            constexpr size_t lds = ci_lds_bytes(8, NW, CI_F16);
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
            GIGA_LAUNCH(kern, dim3(5, 8, B), dim3(NW * 64), lds, s, tsdf, cw, cb, P0, XZP, YZP, B, mask_out);
        }
    };
    pre();
    using TT = std::true_type; using FF = std::false_type;
    if constexpr (SPLIT) run_convin(TT{}, TT{}, FF{});
    else if constexpr (sizeof(T) == 2) run_convin(TT{}, FF{}, FF{});
    else if constexpr (MATH == MATH_BF16) {
        if (bf_ci16 == 2 && keep_mask) run_convin(TT{}, FF{}, TT{});
        else if (bf_ci16) { if (keep_mask) run_convin(TT{}, TT{}, TT{}); else run_convin(TT{}, TT{}, FF{}); }
        else { if (keep_mask) run_convin(FF{}, FF{}, TT{}); else run_convin(FF{}, FF{}, FF{}); }
    } else { if (keep_mask) run_convin(FF{}, FF{}, TT{}); else run_convin(FF{}, FF{}, FF{}); }
    post();
    {
        pre();
        GIGA_LAUNCH(plane_finalize_kernel<T>, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, s, XZP, YZP, P0, B,
                           nxp, reinterpret_cast<unsigned*>(ws + w.SYNC), MEGA_SYNC_WORDS);
        post();
    }
    if (hipGetLastError() != hipSuccess) return -10;

    const int nimg = 3 * B;
    // (tuning knob: GIGA_CONV_XCD=0 keeps the plain workgroup -> weight-group map, see conv_wg_map)
    const int xcd_local = [] { const char* e = getenv("GIGA_CONV_XCD"); return e ? atoi(e) : 1; }();
    // exact fp32 only: the stride-1 3x3 layers whose bit is set run as Winograd F(2x2, 3x3) (giga_wino.h) on their own weight image
    constexpr bool CAN_WINO = sizeof(T) == 4 && MATH == MATH_NATIVE;
    unsigned wino_mask = 0;
    if constexpr (CAN_WINO) {
        for (int l = 0; l < NCONV; ++l)
            if ((kConv[l].kind == CONV3 || kConv[l].kind == UPCONV) && (wino_req >> l & 1u)) wino_mask |= 1u << l;
    }
    auto W_ = [&](int l) {
        if ((wino_mask >> l & 1u) && kConv[l].kind == CONV3) return blob + ko.conv[l].wino;    // (up_run reads the ordinary fp32 fragments)
        return blob + (SPLIT ? ko.conv[l].w16s : MATH == MATH_BF16 ? ko.conv[l].wbf : precision == 1 ? ko.conv[l].w16 : ko.conv[l].w32);
    };
    auto Bi = [&](int l) { return reinterpret_cast<const float*>(blob + ko.conv[l].bias); };
    auto args = [&](int l, const void* i0, const void* i1, void* o, void* op) {
        ConvArgs a{};
        a.in0 = i0; a.in1 = i1; a.w = W_(l); a.bias = Bi(l); a.out = o; a.out_pool = op; a.out_nchw = nullptr;
        a.nimg = nimg;
        a.trace_id = l;
        a.xcd_local = xcd_local;
        return a;
    };
    uint8_t* b = ws;
    int rc = 0;
    // GIGA_FOLD_FINAL: the caller's decoder carries conv_final inside its fc_c weights, so up1.conv2 writes straight
    // into the output planes and the last layer is not launched (its probe stage then brackets nothing)
    void* a6 = fold_final ? planes_nhwc : static_cast<void*>(b + w.A6);
    ConvArgs L[NCONV] = {
        args(0, b + w.P0, nullptr, b + w.A0, nullptr),  args(1, b + w.A0, nullptr, b + w.S0, b + w.Q0),
        args(2, b + w.Q0, nullptr, b + w.A1, nullptr),  args(3, b + w.A1, nullptr, b + w.S1, b + w.Q1),
        args(4, b + w.Q1, nullptr, b + w.A2, nullptr),  args(5, b + w.A2, nullptr, b + w.S2, nullptr),
        args(6, b + w.S2, nullptr, b + w.U0, nullptr),  args(7, b + w.U0, b + w.S1, b + w.A3, nullptr),
        args(8, b + w.A3, nullptr, b + w.A4, nullptr),  args(9, b + w.A4, nullptr, b + w.U1, nullptr),
        args(10, b + w.U1, b + w.S0, b + w.A5, nullptr), args(11, b + w.A5, nullptr, a6, nullptr),
        args(12, b + w.A6, nullptr, planes_nhwc, nullptr)};
    L[12].out_nchw = planes_nchw;
    const int nlayers = fold_final ? 12 : 13;
    // One persistent launch for the whole U-Net (unet_mega_kernel) is the default, unless a single layer is being probed (stages
    // 2..14) -- for every batch size in the f16-class modes, from 24 images (8 scenes) up in fp32 / bf16 (below that an fp32 layer's
    // barrier + first patch cost what its launch costs: 152 vs 156 us at one scene).  The flag GIGA_LAYERWISE_UNET of the call (or
    // GIGA_UNET_PERSIST=0 in the environment, for a whole process) keeps one launch per layer; GIGA_PERSIST_UNET forces the
    // persistent form where the default would not take it.
    static const int env_persist = [] { const char* e = getenv("GIGA_UNET_PERSIST"); return e ? atoi(e) : -1; }();   // -1 unset, 0 off, 1 force
    const bool full_device = [] {                             // per call: the CURRENT device (a process may drive several)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        return cus == 256;                                    // 8 XCDs x 32 CUs, one workgroup per CU
    }();
    constexpr bool F16CLASS = sizeof(T) == 2 || MATH == MATH_SPLIT;
    const bool probe_layer = pr.stage >= 2 && pr.stage <= 14;
    const bool by_default = F16CLASS || nimg >= 24;
    bool mega = full_device && !probe_layer &&
                (persist > 0 || (persist == 0 && (env_persist > 0 || (env_persist < 0 && by_default))));
    int mega_slot = -2;
    if (mega) { mega_slot = persistent_slot(s); mega = mega_slot != -1; }        // at most MEGA_MAX_IN_FLIGHT persistent launches in flight
    // conv32 (giga_conv32.h) or conv16 for the f16-class modes.  A call (GIGA_CONV32_UNET / GIGA_CONV16_UNET) or the process
    // (GIGA_CONV32=1 / 0) can force one; left alone, conv32 runs up to 16 scenes (48 images) and conv16 beyond: conv32's encoder is
    // 3-9 % faster up to 32 scenes, but from ~24 scenes on -- every CU busy -- the chip holds a 3-5 % lower shader clock while and after
    // the conv32 launch runs (at LOWER socket power: a current limit), which costs a sustained 32-scene step up to 4 % and a
    // 128-scene step up to 6 % (tools/gpu_sustained_power.py, profiles/r04/final/sustained_power.txt).
    constexpr int C32MODE = sizeof(T) == 2 ? C32_NATIVE : MATH == MATH_SPLIT ? C32_SPLIT : MATH == MATH_BF16 ? C32_BF16 : -1;
    constexpr int C32_AUTO_MAX_IMAGES = 48;
    static const int env_c32 = [] { const char* e = getenv("GIGA_CONV32"); return e ? (atoi(e) ? 1 : -1) : 0; }();
    if constexpr (C32MODE >= 0) {
        const int want = conv32 != 0 ? conv32 : env_c32 != 0 ? env_c32 : (nimg <= C32_AUTO_MAX_IMAGES ? 1 : -1);
        if (want > 0) {
            auto W32 = [&](int l) { return blob + (C32MODE == C32_SPLIT ? ko.conv[l].c32s : C32MODE == C32_BF16 ? ko.conv[l].c32b : ko.conv[l].c32h); };
            ConvArgs M[NCONV];
            for (int l = 0; l < NCONV; ++l) { M[l] = L[l]; M[l].w = W32(l); M[l].xcd_local = 0; M[l].out_pool = nullptr; }
            M[2].in0 = b + w.S0; M[2].out_pool = b + w.Q0;      // layers 2 and 4 pool their input while staging it
            M[4].in0 = b + w.S1; M[4].out_pool = b + w.Q1;
            if (mega) {
                MegaArgs m{};
                for (int l = 0; l < NCONV; ++l) m.layer[l] = M[l];
                m.sync = reinterpret_cast<unsigned*>(b + w.SYNC);
                m.nlayers = nlayers;
                const unsigned grid = (unsigned)c32_groups(nimg) * MEGA_GROUP;
                // fused same-resolution pairs (C32Pair) while a member's rows fit ONE sub-band of the pair (up to two images per
                // group: the halo rows a member recomputes are 2 of ~12; beyond that the pair needs smaller sub-bands than the
                // single layers and measured slower, profiles/r04/conv32_fused_pairs.txt).  Bit-identical either way;
                // GIGA_C32_FUSE=0 / 1 forces one form (A/B runs).
                static const int env_fuse = [] { const char* e = getenv("GIGA_C32_FUSE"); return e ? atoi(e) : -1; }();
                const int groups = c32_groups(nimg);
                const bool fuse = env_fuse >= 0 ? env_fuse != 0 : (nimg + groups - 1) / groups <= 2;
                stage_no = 15;
                pre();
                auto go = [&](auto kern) {
                    giga::dyn_lds_once(reinterpret_cast<const void*>(kern), C32_LDS_TOTAL);
                    GIGA_LAUNCH(kern, dim3(grid), dim3(C32_NW * 64), C32_LDS_TOTAL, s, m);
                };
                if (fuse) go(unet32_mega_kernel<C32MODE, true>); else go(unet32_mega_kernel<C32MODE, false>);
                g_last_unet_path.store(GIGA_PATH_CONV32 | GIGA_PATH_PERSISTENT | (fuse ? GIGA_PATH_FUSED_PAIRS : 0), std::memory_order_relaxed);
                persistent_launched(mega_slot, s);
                post();
                return hipGetLastError() == hipSuccess ? 0 : -10;
            }
            g_last_unet_path.store(GIGA_PATH_CONV32, std::memory_order_relaxed);
            if (pr.stage == 15) (void)hipEventRecord(pr.ev0, s);
#define X(l, KIND, C0, C1, COUT, H, W, POOLIN, SPW, SGN, SGS, KPS)                                                          \
            if (l < nlayers) { pre(); rc |= launch_conv32<typename U32Layer<C32MODE, l>::G, U32Layer<C32MODE, l>::RELU>(M[l], s); post(); } \
            else { pre(); post(); }
            GIGA_UNET32_LAYERS(X)
#undef X
            if (pr.stage == 15) (void)hipEventRecord(pr.ev1, s);
            return rc;
        }
    }
    g_last_unet_path.store((mega ? GIGA_PATH_PERSISTENT : 0) | (wino_mask ? GIGA_PATH_WINOGRAD : 0), std::memory_order_relaxed);
    if (mega) {
        MegaArgs m{};
        for (int l = 0; l < NCONV; ++l) { m.layer[l] = L[l]; m.layer[l].xcd_local = 0; }   // (the kernel hands every group its images itself)
        m.sync = reinterpret_cast<unsigned*>(b + w.SYNC);
        m.nlayers = nlayers;
        m.wino_mask = wino_mask;
        const int slots = (nimg + 7) / 8 < MEGA_SLOTS ? (nimg + 7) / 8 : MEGA_SLOTS;
        const unsigned grid = 8u * (unsigned)slots * MEGA_GROUP;
        stage_no = 15;                                        // probe stage 15 = the whole U-Net
        auto go = [&](auto kern, const size_t lds, const int nw) {
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
            pre();
            GIGA_LAUNCH(kern, dim3(grid), dim3(nw * 64), lds, s, m);
            persistent_launched(mega_slot, s);
            post();
        };
        constexpr size_t lds = mega_lds_bytes<T, MATH>();
        static_assert(lds <= 160 * 1024, "LDS budget of the persistent U-Net kernel");
        if constexpr (CAN_WINO) {
            constexpr size_t ldsw = mega_lds_bytes<T, MATH, true>();
            static_assert(ldsw <= 160 * 1024, "LDS budget of the persistent U-Net kernel (Winograd stages)");
            if (wino_mask) go(unet_mega_kernel<T, MATH, true>, ldsw, WINO_NW); else go(unet_mega_kernel<T, MATH, false>, lds, MEGA_NW);
        } else {
            go(unet_mega_kernel<T, MATH, false>, lds, MEGA_NW);
        }
        return hipGetLastError() == hipSuccess ? 0 : -10;
    }
    if (pr.stage == 15) (void)hipEventRecord(pr.ev0, s);
#define X(l, KIND, C0, C1, COUT, H, W, NB, POOL, NB16)                                                                    \
    if (l < nlayers) {                                                                                                    \
        pre();                                                                                                            \
        bool direct = true;                                                                                               \
        if constexpr (CAN_WINO && KIND == CONV3) {                                                                        \
            if (wino_mask >> l & 1u) { direct = false; rc |= launch_wino<C0, C1, COUT, H, W, POOL>(L[l], s); }            \
        }                                                                                                                 \
        if constexpr (CAN_WINO && KIND == UPCONV) {                                                                       \
            if (wino_mask >> l & 1u) { direct = false; rc |= launch_up<C0, COUT, H, W>(L[l], s); }                        \
        }                                                                                                                 \
        if (direct) rc |= launch_conv<T, KIND, C0, C1, COUT, H, W, (sizeof(T) == 2 ? NB16 : NB), POOL, KIND == CONV3, MATH>(L[l], s); \
        post();                                                                                                           \
    }                                                                                                                     \
    else { pre(); post(); }
    GIGA_UNET_LAYERS(X)
#undef X
    if (pr.stage == 15) (void)hipEventRecord(pr.ev1, s);
    return rc;
}

int launch_encoder(const float* tsdf, const uint8_t* blob, void* planes_nhwc, float* planes_nchw, int B,
                   int precision, uint8_t* ws, hipStream_t s, int probe_stage, void* ev0, void* ev1) {
    if (B <= 0) return 0;
    Probe pr{ev0 && ev1 ? probe_stage : -1, static_cast<hipEvent_t>(ev0), static_cast<hipEvent_t>(ev1)};
    const bool fold = (precision & GIGA_FOLD_FINAL) != 0;
    const int persist = (precision & GIGA_LAYERWISE_UNET) ? -1 : (precision & GIGA_PERSIST_UNET) ? 1 : 0;   // -1 per-layer launches, 0 auto, 1 persistent
    const int prec = precision & ~(GIGA_FOLD_FINAL | GIGA_PERSIST_UNET | GIGA_LAYERWISE_UNET | GIGA_CONV32_UNET | GIGA_CONV16_UNET | GIGA_CONVIN_MASK | GIGA_DIRECT_CONV);
    const int c32 = (precision & GIGA_CONV32_UNET) ? 1 : (precision & GIGA_CONV16_UNET) ? -1 : 0;      // 1 conv32, -1 conv16, 0 default
    const bool km = (precision & GIGA_CONVIN_MASK) != 0;      // training forward (precisions 0 and 3): keep conv_in's ReLU mask for the backward
    // Winograd F(2x2, 3x3) for the 3x3 layers of precision 0 (giga_wino.h): on unless the call says GIGA_DIRECT_CONV (a blob rebuilt on
    // the device needs giga_derive_winograd first); GIGA_WINOGRAD=<mask> in the environment selects the layers of a whole process
    // (0 = none; A/B runs)
    static const unsigned env_wino = [] { const char* e = getenv("GIGA_WINOGRAD"); return e ? (unsigned)strtoul(e, nullptr, 0) : WINO_DEFAULT_MASK; }();
    const unsigned wino = (precision & GIGA_DIRECT_CONV) ? 0u : env_wino;
    if (prec == 2) return encoder_run<float, MATH_SPLIT>(tsdf, blob, planes_nhwc, planes_nchw, B, ws, s, pr, fold, persist, c32);
    if (prec == 3) return encoder_run<float, MATH_BF16>(tsdf, blob, planes_nhwc, planes_nchw, B, ws, s, pr, fold, persist, c32, km);
    return prec == 1 ? encoder_run<half_t>(tsdf, blob, planes_nhwc, planes_nchw, B, ws, s, pr, fold, persist, c32)
                     : encoder_run<float>(tsdf, blob, planes_nhwc, planes_nchw, B, ws, s, pr, fold, persist, c32, km, wino);
}

}  // namespace giga

extern "C" int giga_encoder_last_path(void) { return giga::g_last_unet_path.load(std::memory_order_relaxed); }

#ifdef GIGA_TRACE
// diagnostic build only: select the traced U-Net layer (host_out == nullptr) or read the timeline back
extern "C" int giga_debug_mega_trace(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_mega_trace), sizeof(long long) * 8 * 32) == hipSuccess ? 0 : -10;
}
extern "C" int giga_debug_c32_trace(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_c32_trace), sizeof(long long) * giga::NCONV * giga::C32_NW * 8) == hipSuccess ? 0 : -10;
}
extern "C" int giga_debug_convin_trace(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_ci_trace), sizeof(long long) * 8 * 64) == hipSuccess ? 0 : -10;
}
extern "C" int giga_debug_conv_trace(int layer, long long* host_out) {
    if (!host_out) {                                  // select the layer and clear the previous layer's timeline
        static const long long zeros[giga::CONV_NW * 64] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(giga::g_conv_trace), zeros, sizeof(zeros)) != hipSuccess) return -10;
        return hipMemcpyToSymbol(HIP_SYMBOL(giga::g_conv_trace_layer), &layer, sizeof(int)) == hipSuccess ? 0 : -10;
    }
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_conv_trace), sizeof(long long) * giga::CONV_NW * 64) == hipSuccess ? 0 : -10;
}
#endif
