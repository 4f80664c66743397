// wino: the stride-1 3x3 U-Net layers of the exact-fp32 path as Winograd F(2x2, 3x3) on the fp32 MFMA.
//
//   reference: encoder/unet.py:14-23 (conv3x3, padding 1), :48-72 (DownConv: conv1, conv2, pool), :101-114 (UpConv conv1/conv2)
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      d = 4x4 input patch of a 2x2 output tile, g = 3x3 kernel
//   16 multiplies per (tile, cin, cout) instead of 36: the direct form (giga_conv16.h) sits at 0.41-0.73 of the fp32-MFMA peak and
//   cannot go above 1; this form needs 0.44 of its MFMA instructions.  fp32-input MFMA does not co-execute with the VALU (DESIGN
//   "fp32 MFMA and the VALU"), so both transforms are paid in MFMA time -- they are laid out so that they cost ~15 %:
//     * GEMM per Winograd position p = 4*xi + nu:  M[p][co][tile] = sum_ci U[p][ci][co] * V[p][ci][tile]  on v_mfma_f32_16x16x4_f32,
//       weights = A operand (rows = 16 output channels), transformed patches = B operand (columns = 16 tiles), K = 4 input channels.
//       Lane (j = lane & 15, g = lane >> 4) supplies B[k = g][col = j]: the lane OWNS tile j and channels 4g .. 4g+3 of every 16-channel
//       chunk, reads that tile's 4x4 patch from LDS itself and transforms it in registers -- the B operand never moves across lanes.
//     * D row = 4g + r: a lane's four registers of position p are four consecutive output channels of ITS tile, so A^T M A is
//       lane-local too, the 2x2 max-pool is a max over the lane's four results, and every store is one 16-byte vector.
//     * both transforms are written on 2- / 4-float vectors (v_pk_add_f32): 32 packed adds per 32 MFMAs on the way in, ~50 per unit
//       (256 MFMAs per 32 input channels) on the way out.
//   Unit = (block of BW x BH tiles <= 16, 16 output channels).  Wave-independent like conv16: a workgroup keeps the Winograd-domain
//   weights of its 16-channel group resident in LDS (16 positions x CIN x 16 x 4 B, one LDS-DMA fill), every wave walks its own
//   units (dealt round-robin inside the workgroup): haloed (2BH+2) x (2BW+2) patch of 16 input channels -> wave-private LDS (pixel
//   stride 80 B, row stride 832 / 1040 B), loads two chunks ahead of the arithmetic, no workgroup barrier.
//   Inputs wider than 64 channels run as CIN/64 K-PASSES: pass k keeps the weights of channels 64k .. 64k+63 resident, and a wave adds
//   the pre-activation partials the previous pass left in the output (written by a wave of its own workgroup, behind a barrier)
//   before bias / ReLU.
//   The same kernel is the DATA-GRADIENT convolution of a 3x3 layer in the fp32 training step (RELU = false: no bias, the ReLU mask of
//   the layer below in the epilogue; images of the flipped / transposed weights in the backward blob).
//   fp32 Winograd is not the bitwise fma chain of the direct form: results differ from it at the 1e-6 level (tests hold 1e-4 to the
//   oracle); GIGA_WINOGRAD=0 / the GIGA_DIRECT_CONV flag (forward) and GIGA_WINOGRAD_BWD=0 (data gradients) keep the direct kernels.
//   A blob rebuilt on the device (training: giga_repack_device is a pure gather of parameters, the Winograd image is not) gets its
//   images from giga_derive_winograd, bit-identical to the host packer's.
#pragma once
#include "giga_conv16.h"

namespace giga {

// layers that run as Winograd by default (bit l = U-Net layer l of giga_layout.h::kConv); settled by measurement
constexpr unsigned WINO_DEFAULT_MASK = 0xFFF;     // all ten 3x3 layers (0, 1, 2, 3, 4, 5, 7, 8, 10, 11) as Winograd, the two ConvTranspose layers
                                                  // (6, 9) as plain GEMMs on the same lane layout (up_run); profiles/r06/wino_ab_*.txt
constexpr int WINO_NW = 8;                        // waves per workgroup: two per SIMD, 256 registers each (64 accumulators + 32 transformed
                                                  // values + a prefetched chunk do not fit the 168 of three per SIMD)
constexpr int WINO_PS = 80;                       // LDS pixel stride: 16 channels x 4 B + 16 B pad
template <int H, int W> struct WinoBlock {        // tiles per unit: 4 x 4 where the 20 x 20 tile grid of a 40^2 image divides, else 5 x 3
    static constexpr int TW = W / 2, TH = H / 2;
    static constexpr int BW = TW % 4 == 0 ? 4 : 5, BH = TW % 4 == 0 ? 4 : 3;
    static constexpr int TXB = (TW + BW - 1) / BW, TYB = (TH + BH - 1) / BH;       // blocks per image
    static constexpr int PW = 2 * BW + 2, PH = 2 * BH + 2;                         // staged patch (pixels)
    static constexpr int RS = BW == 4 ? 832 : 1040;                                // LDS row stride (bytes): 52 / 65 sixteen-byte slots
    static constexpr int REGION = PH * RS;                                         // 8320 B either way
    static_assert(BW * BH <= 16 && PW * WINO_PS <= RS, "tile block");
};
constexpr int wino_kpass(int cin) { return cin > 64 ? cin / 64 : 1; }
template <int C0, int C1, int H, int W>
constexpr int wino_nw() {                         // waves per workgroup: WINO_NW unless weights + patches would not fit the 160 KiB LDS
    constexpr int CINP = (C0 + C1) / wino_kpass(C0 + C1);
    constexpr int fit = (160 * 1024 - 16 * CINP * 16 * 4) / WinoBlock<H, W>::REGION;
    return fit >= WINO_NW ? WINO_NW : (fit / 2) * 2;
}
template <int C0, int C1, int H, int W>
constexpr size_t wino_lds_bytes() {
    constexpr int CINP = (C0 + C1) / wino_kpass(C0 + C1);
    return (size_t)16 * CINP * 16 * 4 + (size_t)wino_nw<C0, C1, H, W>() * WinoBlock<H, W>::REGION;
}
// Winograd weight image of a layer (giga_pack.cpp::pack_wino): [grp = cout / 16][kpass][pos 16][chunk of 16 ci][half 2][lane 64][2 floats]
//   value = U[pos][ci = 64 * kpass + 16 * chunk + 4 * (lane >> 4) + 2 * half + e][co = 16 * grp + (lane & 15)],  U = G g G^T
constexpr size_t wino_image_bytes(int cin, int cout) { return (size_t)16 * cin * cout * 4; }

// 16-byte store at uniform base + 32-bit byte offset.  Plain C++ on purpose: the inline-asm saddr form of giga_dev.h is only safe for
// <= 64-bit data -- a wider VMEM store reads its data registers over several cycles, the hazard recogniser does not see into the asm,
// and the compiler reuses the registers at once (measured: the un-pooled layers stored garbage in two of four channels).
__device__ __forceinline__ void store_f32x4(float* uniform_base, unsigned byte_off, f32x4v v) {
    *reinterpret_cast<f32x4v*>(reinterpret_cast<char*>(uniform_base) + byte_off) = v;
}

template <int C0, int C1, int COUT>
__device__ __forceinline__ void wino_fill(const ConvArgs& a, uint8_t* smem, int block, int nblocks, int kpass) {
    constexpr int CIN = C0 + C1, KP = wino_kpass(CIN), CINP = CIN / KP, NGRP = COUT / 16;
    constexpr int WFRAGS = 16 * (CINP / 16);                     // 1 KiB pieces of one (group, pass)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwl = blockDim.x >> 6;
    const int grp = conv_wg_map<NGRP>(a, block, nblocks).grp;
    const uint8_t* wsrc = a.w + ((size_t)grp * KP + kpass) * WFRAGS * FRAG;
    for (int c = wave; c < WFRAGS; c += nwl)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wsrc + (size_t)c * FRAG + lane * 16),
            (__attribute__((address_space(3))) void*)(smem + c * FRAG), 16, 0, 0);
}

// one K-pass of the layer for this workgroup's units.  KPASS_FIRST / KPASS_LAST: the first pass starts from zero, later ones add the
// partial the previous pass left in `out`; only the last applies bias, ReLU and the pool.
template <int C0, int C1, int COUT, int H, int W, bool POOL, bool RELU>
__device__ __forceinline__ void wino_run(const ConvArgs& a, uint8_t* smem, int block, int nblocks, int kpass) {
    using G = WinoBlock<H, W>;
    constexpr int CIN = C0 + C1, KP = wino_kpass(CIN), CINP = CIN / KP, NCHUNK = CINP / 16, NGRP = COUT / 16;
    constexpr int NWV = wino_nw<C0, C1, H, W>();
    constexpr int BW = G::BW, BH = G::BH, PW = G::PW, PH = G::PH, RS = G::RS, PS = WINO_PS, REGION = G::REGION;
    constexpr int NVEC = PW * PH * 4, NLD = (NVEC + 63) / 64;   // 16-byte vectors of a staged chunk, staging loads per lane
    constexpr int WBYTES = 16 * CINP * 16 * 4;
    static_assert(C1 == 0 || (C0 % 16 == 0 && (KP == 1 || C0 % CINP == 0)), "concat boundary on a chunk / pass boundary");
    static_assert(!POOL || KP == 1, "pooled layers are single-pass");
    static_assert(KP <= 2, "a third pass would read partials back through an L1 line it has read before");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const bool active = wave < NWV;
    CONV_T(0);
    uint8_t* region = smem + WBYTES + (active ? wave : 0) * REGION;
    const ConvWgMap wm = conv_wg_map<NGRP>(a, block, nblocks);
    const int grp = wm.grp, wg_in_grp = wm.wg_in_grp, wgs_per_grp = wm.wgs_per_grp;
    // A workgroup owns a contiguous, balanced range [ulo, uhi) of its weight group's units and deals them round-robin over its waves:
    // wave w takes ulo + w, ulo + w + 8, ...  Waves w and w + 4 share a SIMD (a workgroup's waves go to the SIMDs cyclically), and the
    // two waves of a SIMD share its matrix pipe, so what counts is units per SIMD: round-robin gives 5 / 5 / 5 / 4 for 19 units and
    // 3 / 3 / 3 / 3 for 12 -- the optimum.  (Measured equal: units drawn from an LDS counter, 230.4 vs 229.9 us per U-Net launch; a stride
    // over ALL workgroups' waves, conv16's, puts every remainder on the first waves of each workgroup: 6 / 5.5 / 5 / 5.)
    const int units_all = wm.nimg * G::TYB * G::TXB;
    const int ulo = (int)((long long)units_all * wg_in_grp / wgs_per_grp), uhi = (int)((long long)units_all * (wg_in_grp + 1) / wgs_per_grp);
    const int units = uhi;
    const bool klast = kpass == KP - 1;

    // this lane's tile inside the block (lanes beyond the block's tiles shadow tile 0 and store nothing)
    const int jt = j < BW * BH ? j : 0;
    const int tyl = jt / BW, txl = jt % BW;
    const int a_off = 2 * tyl * RS + 2 * txl * PS + 16 * g;

    // Staging geometry of this lane's NLD vectors: vector i = lane + 64 q is 16-byte vector (i & 3) = (lane & 3) of patch pixel i >> 2.
    // fp32 MFMA and VALU do not overlap, so the staging is kept free of VALU work: the sources are read with BUFFER loads -- a vector
    // outside the image (or beyond the patch) carries an offset beyond the buffer's size and comes back as zeros, so every load and
    // every ds_write is unconditional and nobody zero-fills; a chunk's loads are `per-unit offset register + chunk offset in an SGPR`.
    const int v16 = (lane & 3) * 16;
    const uint32_t rowb = (uint32_t)(a.cs0 ? a.cs0 : C0) * 4;          // bytes per pixel of the source tensors (launch_wino: both alike)
    constexpr uint32_t OOB = 0x80000000u;             // beyond every source tensor (launch_wino: < 2 GiB) whether or not the hardware's range
                                                      // check adds the SGPR chunk offset, and far from wrapping when it is added
    int st_lds[NLD];
    uint32_t st_rel[NLD];                              // byte offset of the vector from the patch origin's pixel, in the source tensor
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int i = lane + 64 * q, pix = i >> 2, ly = pix / PW, lx = pix % PW;
        st_lds[q] = i < NVEC ? ly * RS + lx * PS + v16 : 64;           // (beyond the patch: the pad bytes of pixel 0)
        st_rel[q] = i < NVEC ? (uint32_t)(ly * W + lx) * rowb + (uint32_t)v16 : OOB;
    }
    const uint32_t in_bytes = (uint32_t)a.nimg * H * W * rowb;         // (conv_image_range has moved the bases to this group's first image)
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in0)) + (size_t)a.co0 * 4, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = C1 > 0 ? __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in1)) + (size_t)a.co1 * 4, 0, in_bytes, 0x00020000) : rs0;
    auto unit_coords = [&](int u, int& bx, int& by, int& img) {       // (u is wave-uniform: scalar arithmetic)
        bx = u % G::TXB; by = (u / G::TXB) % G::TYB; img = wm.img0 + u / (G::TXB * G::TYB);
    };
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v stg[NLD];
    uint32_t st_off[NLD];                              // this unit's byte offsets (OOB outside the image)
    auto unit_setup = [&](int u) {
        int bx, by, img;
        unit_coords(u, bx, by, img);
        const int y0 = 2 * BH * by - 1, x0 = 2 * BW * bx - 1;
        const uint32_t base = (uint32_t)((img * H + y0) * W + x0) * rowb;          // wraps for y0 = -1: only added to in-image vectors (mod 2^32)
        if (y0 >= 0 && x0 >= 0 && y0 + PH <= H && x0 + PW <= W) {       // interior block (uniform): no bounds tests
#pragma unroll
            for (int q = 0; q < NLD; ++q) st_off[q] = (64 * (q + 1) > NVEC && st_rel[q] == OOB) ? OOB : st_rel[q] + base;
        } else {
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int pix = (lane >> 2) + 16 * q;
                const int ly = (pix * (65536 / PW + 1)) >> 16, lx = pix - ly * PW;          // pix / PW for pix < 2^8
                const bool ok = st_rel[q] != OOB && (unsigned)(y0 + ly) < (unsigned)H && (unsigned)(x0 + lx) < (unsigned)W;
                st_off[q] = ok ? st_rel[q] + base : OOB;
            }
        }
    };
    auto issue_loads = [&](int cc) {                   // cc = chunk inside this pass (compile-time after unrolling)
        const int c16 = kpass * CINP + cc * 16;        // first channel of the chunk in the concatenated input
        const bool first = C1 == 0 || c16 < C0;
        const int soff = (first ? c16 : c16 - C0) * 4;
#pragma unroll
        for (int q = 0; q < NLD; ++q)
            stg[q] = first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, st_off[q], soff, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(rs1, st_off[q], soff, 0);
    };

    int u = ulo + wave;
    const bool work = active && u < units;
    if (work) { unit_setup(u); issue_loads(0); }
    const f32x4v bias4 = (a.bias && klast) ? *reinterpret_cast<const f32x4v*>(a.bias + grp * 16 + 4 * g) : f32x4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): weights (LDS-DMA), first patch chunk and bias landed
    __syncthreads();
    if (!work) return;
    CONV_T(1);
    int tcount = 2;                                    // (diagnostic builds: 2 + 2k = unit k's first burst issued, 3 + 2k = its epilogue done)

    f32x4v acc[16];
    const uint8_t* wl = smem + lane * 8;               // this lane's 8 bytes of a (position, chunk, half) piece
    float* outp = reinterpret_cast<float*>(a.out);

    auto epilogue = [&](const int u) {
        int bx, by, img;
        unit_coords(u, bx, by, img);
        const int ty = BH * by + tyl, tx = BW * bx + txl;
        const bool ok = j < BW * BH && ty < G::TH && tx < G::TW;
        // A^T M A on four-channel vectors (a - (b + c): the form whose additions the compiler packs)
        f32x4v t0[4], t1[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            t0[xi] = (acc[4 * xi] + acc[4 * xi + 1]) + acc[4 * xi + 2];
            t1[xi] = acc[4 * xi + 1] - (acc[4 * xi + 2] + acc[4 * xi + 3]);
        }
        f32x4v y[4];
        y[0] = (t0[0] + t0[1]) + t0[2]; y[1] = (t1[0] + t1[1]) + t1[2];
        y[2] = t0[1] - (t0[2] + t0[3]); y[3] = t1[1] - (t1[2] + t1[3]);
        const uint32_t pix = (uint32_t)((img * H + 2 * ty) * W + 2 * tx);
        const uint32_t o00 = (pix * COUT + grp * 16 + 4 * g) * 4u;
        if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t off = o00 + (uint32_t)(((e >> 1) * W + (e & 1)) * COUT * 4);
                if constexpr (KP > 1) {
                    if (kpass > 0) y[e] += *reinterpret_cast<const f32x4v*>(reinterpret_cast<const char*>(outp) + off);
                }
                if (klast) {
                    if (RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[e][r] = relu(y[e][r]);
                    } else if (a.mask) {             // data-gradient convolutions: the ReLU backward of the layer below, fused (ConvArgs::mask)
                        const f32x4v mk = *reinterpret_cast<const f32x4v*>(reinterpret_cast<const char*>(a.mask) + off);
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[e][r] = mk[r] > 0.f ? y[e][r] : 0.f;
                    }
                }
                store_f32x4(outp, off, y[e]);
            }
            if constexpr (POOL) {
                f32x4v mx;
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[r] = fmaxf(fmaxf(y[0][r], y[1][r]), fmaxf(y[2][r], y[3][r]));
                const uint32_t pp = (uint32_t)((img * (H / 2) + ty) * (W / 2) + tx);
                store_f32x4(reinterpret_cast<float*>(a.out_pool), (pp * COUT + grp * 16 + 4 * g) * 4u, mx);
            }
        }
    };

    // ---- the wave's instruction stream -------------------------------------------------------------------------------------------
    // A half-chunk (this lane's channels 4g + 2h, 4g + 2h + 1 of a 16-channel chunk) is: 16 ds_read_b64 of the raw patch -> 32 packed
    // adds (B^T d B) -> a burst of 32 MFMAs.  Only the adds have to sit between two bursts; everything else rides INSIDE a burst, after
    // its fourth MFMA pair: the raw reads of the NEXT half, and at a chunk boundary the ds_writes that stage the next chunk and the
    // buffer loads of the chunk after that (loads run two chunks ahead of the arithmetic).
    f32x2v d[4][4];
    auto load_d = [&](const int h) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int x = 0; x < 4; ++x)
                // (volatile LDS access on purpose: left alone, the compiler pairs these into ds_read2_b64, which the LDS serves as 16-lane
                //  groups over 32 banks -- the four tile rows of a block then collide 4-way, 192 extra LDS cycles per half-chunk, the LDS
                //  saturated (52.7 M conflict cycles per U-Net launch, profiles/r06/final/pmc.txt).  A plain ds_read_b64 is served as two
                //  32-lane groups over 64 banks: 2-way here, which tile stride 2 makes the floor for 8-byte reads of this layout)
                d[r][x] = *(const volatile __attribute__((address_space(3))) f32x2v*)(region + a_off + r * RS + x * PS + 8 * h);
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) *reinterpret_cast<u32x4v*>(region + st_lds[q]) = stg[q];
    };
    f32x2v v[16];
    auto transform = [&]() {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x2v e0 = d[0][x] - d[2][x], e1 = d[1][x] + d[2][x], e2 = d[2][x] - d[1][x], e3 = d[1][x] - d[3][x];
            d[0][x] = e0; d[1][x] = e1; d[2][x] = e2; d[3][x] = e3;
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            v[4 * xi + 0] = d[xi][0] - d[xi][2];
            v[4 * xi + 1] = d[xi][1] + d[xi][2];
            v[4 * xi + 2] = d[xi][2] - d[xi][1];
            v[4 * xi + 3] = d[xi][1] - d[xi][3];
        }
    };
    // 32 MFMAs: positions in pairs (two independent accumulators alternate: 40-cycle dependent latency, 32-cycle issue); the weights
    // of pair pp + PD are read from LDS right after the MFMAs of pair pp are issued; `mid` runs after pair 3
    auto burst = [&](auto first_tag, const int cc, const int h, auto&& mid) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint8_t* wbase = wl + (cc * 2 + h) * 512;              // [pos][chunk][half][lane][2]: pos stride = NCHUNK KiB
        constexpr int PD = 4;
        f32x2v wq[PD][2];
#pragma unroll
        for (int pp = 0; pp < PD; ++pp) {
            wq[pp][0] = *reinterpret_cast<const f32x2v*>(wbase + (2 * pp) * NCHUNK * 1024);
            wq[pp][1] = *reinterpret_cast<const f32x2v*>(wbase + (2 * pp + 1) * NCHUNK * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int sl = pp % PD, p0 = 2 * pp, p1 = 2 * pp + 1;
            if constexpr (FIRST) {
                // (position 5 = (xi 1, nu 1) enters all four outputs of A^T M A with weight +1: the bias rides in its C operand)
                acc[p0] = mfma32_16(wq[sl][0][0], v[p0][0], f32x4v{0.f, 0.f, 0.f, 0.f});
                acc[p1] = mfma32_16(wq[sl][1][0], v[p1][0], p1 == 5 ? bias4 : f32x4v{0.f, 0.f, 0.f, 0.f});
            } else {
                acc[p0] = mfma32_16(wq[sl][0][0], v[p0][0], acc[p0]);
                acc[p1] = mfma32_16(wq[sl][1][0], v[p1][0], acc[p1]);
            }
            acc[p0] = mfma32_16(wq[sl][0][1], v[p0][1], acc[p0]);
            acc[p1] = mfma32_16(wq[sl][1][1], v[p1][1], acc[p1]);
            __builtin_amdgcn_sched_barrier(0);
            if (pp + PD < 8) {
                wq[sl][0] = *reinterpret_cast<const f32x2v*>(wbase + (2 * (pp + PD)) * NCHUNK * 1024);
                wq[sl][1] = *reinterpret_cast<const f32x2v*>(wbase + (2 * (pp + PD) + 1) * NCHUNK * 1024);
            }
            if (pp == 3) mid();
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    static_assert(NCHUNK >= 2, "loads run two chunks ahead");
    // prologue: chunk 0 of the first unit is in the staging registers (requested before the weight barrier)
    stage_write();
    issue_loads(1);
    load_d(0);
    // unit loop (runtime, uniform); the chunk loop inside is unrolled: what rides in which burst is known at compile time
    while (true) {
        const int un = u + NWV;
        const bool more = un < units;
#pragma unroll
        for (int cc = 0; cc < NCHUNK; ++cc) {
            transform();                                              // (cc, 0): raw values were read inside the previous burst
            auto mid0 = [&]() { load_d(1); };
            if (cc == 0) burst(std::true_type{}, cc, 0, mid0); else burst(std::false_type{}, cc, 0, mid0);
            if (cc == 0) { CONV_T(tcount); ++tcount; }
            transform();                                              // (cc, 1)
            // second burst of the chunk: stage the chunk after it, request the chunk after that, read the first half of the next chunk
            auto mid1 = [&]() {
                if (cc + 1 < NCHUNK) {                                // next chunk of this unit
                    stage_write();
                    if (cc + 2 < NCHUNK) issue_loads(cc + 2);
                    else if (more) { unit_setup(un); issue_loads(0); }
                    load_d(0);
                } else if (more) {                                    // chunk 0 of the next unit (st_off already describes that unit)
                    stage_write();
                    issue_loads(1);
                    load_d(0);
                }
            };
            burst(std::false_type{}, cc, 1, mid1);
        }
        epilogue(u);
        CONV_T(tcount); ++tcount;
        if (!more) break;
        u = un;
    }
    CONV_T(63);
}

template <int C0, int C1, int COUT, int H, int W, bool POOL, bool RELU>
__global__ __launch_bounds__((wino_nw<C0, C1, H, W>() * 64)) void wino_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KP = wino_kpass(C0 + C1);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k > 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);       // this wave's partial stores are acknowledged before it reads them back
            __syncthreads();                           // everyone has left the previous pass's weights
        }
        wino_fill<C0, C1, COUT>(a, smem, (int)blockIdx.x, (int)gridDim.x, k);
        wino_run<C0, C1, COUT, H, W, POOL, RELU>(a, smem, (int)blockIdx.x, (int)gridDim.x, k);
    }
}

// ----------------------------------------------------------------------------------------------------------------------------------
// up: ConvTranspose2d(k = 2, s = 2) of the exact-fp32 path (encoder/unet.py:25-31,101-104: no activation) as four plain GEMMs on the
// same lane layout -- out[2y + dy][2x + dx][co] = bias[co] + sum_ci in[y][x][ci] W[ci][co][dy][dx].  Unit = (16 consecutive pixels of the
// workgroup's image range, 16 output channels); lane (j, g) reads channels 4g .. 4g+3 of every 16-channel chunk of ITS pixel straight
// from memory into the B operand (no LDS staging, no transform), the four sub-pixel products accumulate side by side (four independent
// accumulators: no dependent-MFMA stalls), a lane's D registers are four consecutive channels of its pixel: four 16-byte stores.  The
// A operands are the layer's ordinary fp32 conv16 fragments (same lane layout; a pure gather of parameters, so the training path's
// device repack keeps them valid), the four sub-pixel runs of the workgroup's channel block resident in LDS (CIN x 256 B).
// The direct form (conv16 UPCONV) stages a patch per 32-channel chunk and is latency-bound: 0.27-0.30 of the fp32 peak, and 16 / 14 us
// per layer inside the 8-wave persistent launch (profiles/r06/unet_trace_wino_8waves.txt).
// ----------------------------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__device__ __forceinline__ void up_fill(const ConvArgs& a, uint8_t* smem, int block, int nblocks) {
    constexpr int KG = CIN / 16, NBT = COUT / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwl = blockDim.x >> 6;
    const int grp = conv_wg_map<NBT>(a, block, nblocks).grp;
    for (int c = wave; c < 4 * KG; c += nwl) {                    // piece c = (sub, kg): conv16 fragment (sub * NBT + grp) * KG + kg
        const int sub = c / KG, kg = c - sub * KG;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a.w + ((size_t)(sub * NBT + grp) * KG + kg) * FRAG + lane * 16),
            (__attribute__((address_space(3))) void*)(smem + c * FRAG), 16, 0, 0);
    }
}
template <int CIN, int COUT, int H, int W>
__device__ __forceinline__ void up_run(const ConvArgs& a, uint8_t* smem, int block, int nblocks) {
    constexpr int KG = CIN / 16, NBT = COUT / 16, NWV = WINO_NW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const bool active = wave < NWV;
    const ConvWgMap wm = conv_wg_map<NBT>(a, block, nblocks);
    const int grp = wm.grp;
    const int npix = wm.nimg * H * W, p0 = wm.img0 * H * W;      // this workgroup's pixel range of the input tensor
    const int units_all = (npix + 15) / 16;
    const int ulo = (int)((long long)units_all * wm.wg_in_grp / wm.wgs_per_grp), uhi = (int)((long long)units_all * (wm.wg_in_grp + 1) / wm.wgs_per_grp);
    const uint32_t rowb = (uint32_t)(a.cs0 ? a.cs0 : CIN) * 4;
    const char* in = reinterpret_cast<const char*>(a.in0) + (size_t)a.co0 * 4 + 16 * g;
    f32x4v x[2][KG];                                              // B operands of this and of the next unit
    auto issue = [&](int u, f32x4v (&dst)[KG]) {
        int p = 16 * u + j;
        p = p < npix ? p : npix - 1;
        const char* q = in + (size_t)(p0 + p) * rowb;
#pragma unroll
        for (int k = 0; k < KG; ++k) dst[k] = *reinterpret_cast<const f32x4v*>(q + 64 * k);
    };
    int u = ulo + wave;
    const bool work = active && u < uhi;
    if (work) issue(u, x[0]);
    const f32x4v bias4 = a.bias ? *reinterpret_cast<const f32x4v*>(a.bias + grp * 16 + 4 * g) : f32x4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (!work) return;
    const f32x4v* wl = reinterpret_cast<const f32x4v*>(smem) + lane;
    float* outp = reinterpret_cast<float*>(a.out);
    auto unit = [&](const int u, const f32x4v (&xb)[KG]) {
        f32x4v acc[4];
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            f32x4v wv[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) wv[sub] = wl[(sub * KG + k) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
                    acc[sub] = (k == 0 && e == 0) ? mfma32_16(wv[sub][e], xb[k][e], bias4) : mfma32_16(wv[sub][e], xb[k][e], acc[sub]);
        }
        const int p = 16 * u + j;
        if (p < npix) {
            const int pg = p0 + p, img = pg / (H * W), r = pg - img * (H * W), y = r / W, xx = r - y * W;
            const uint32_t o00 = ((uint32_t)((img * 2 * H + 2 * y) * 2 * W + 2 * xx) * COUT + grp * 16 + 4 * g) * 4u;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
                store_f32x4(outp, o00 + (uint32_t)(((sub >> 1) * 2 * W + (sub & 1)) * COUT * 4), acc[sub]);
        }
    };
    while (true) {                                                 // units dealt round-robin over the waves (see wino_run), two in flight
        const int un = u + NWV;
        const bool more = un < uhi;
        if (more) issue(un, x[1]);
        unit(u, x[0]);
        if (!more) break;
        const int un2 = un + NWV;
        const bool more2 = un2 < uhi;
        if (more2) issue(un2, x[0]);
        unit(un, x[1]);
        if (!more2) break;
        u = un2;
    }
}
template <int CIN, int COUT>
constexpr size_t up_lds_bytes() { return (size_t)4 * (CIN / 16) * FRAG; }
template <int CIN, int COUT, int H, int W>
__global__ __launch_bounds__(WINO_NW * 64) void up_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    up_fill<CIN, COUT>(a, smem, (int)blockIdx.x, (int)gridDim.x);
    up_run<CIN, COUT, H, W>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}
template <int CIN, int COUT, int H, int W>
inline int launch_up(const ConvArgs& a, hipStream_t s) {
    constexpr int NBT = COUT / 16;
    constexpr size_t lds = up_lds_bytes<CIN, COUT>();
    const size_t in_pix = (size_t)a.nimg * H * W;
    if (in_pix * 4 >= (1u << 24) || in_pix * 4 * COUT * 4 >= (1ull << 32)) return -7;
    const int units = (int)((in_pix + 15) / 16);
    int wgs = (units + WINO_NW - 1) / WINO_NW;
    if (wgs > 256 / NBT) wgs = 256 / NBT;
    else if (a.xcd_local && a.nimg % 8 == 0 && wgs % 8 != 0 && wgs + 8 - wgs % 8 <= 256 / NBT) wgs += 8 - wgs % 8;
    auto kern = up_kernel<CIN, COUT, H, W>;
    if (lds > 48 * 1024) giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
    GIGA_LAUNCH(kern, dim3(wgs * NBT), dim3(WINO_NW * 64), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

template <int C0, int C1, int COUT, int H, int W, bool POOL, bool RELU = true>
inline int launch_wino(const ConvArgs& a, hipStream_t s) {
    using G = WinoBlock<H, W>;
    constexpr int NWV = wino_nw<C0, C1, H, W>(), NGRP = COUT / 16;
    constexpr size_t lds = wino_lds_bytes<C0, C1, H, W>();
    static_assert(lds <= 160 * 1024 && NWV >= 4, "LDS budget");
    static_assert(256 % NGRP == 0, "weight groups must divide the CU count");
    const size_t in_pix = (size_t)a.nimg * H * W;
    if (C1 > 0 && (a.cs0 ? a.cs0 : C0) != (a.cs1 ? a.cs1 : C1)) return -7;              // one pixel stride for both sources (wino_run)
    if (in_pix * (size_t)((a.cs0 ? a.cs0 : C0) * 4) >= (1ull << 31)) return -7;           // the buffer loads' out-of-range sentinel is 2 GiB
    if (in_pix >= (1u << 24) || in_pix * (size_t)((a.cs0 ? a.cs0 : C0) * 4) >= (1ull << 32) ||
        (C1 > 0 && in_pix * (size_t)((a.cs1 ? a.cs1 : C1) * 4) >= (1ull << 32)) || in_pix * COUT * 4 >= (1ull << 32)) return -7;
    const int units = a.nimg * G::TXB * G::TYB;
    int wgs = (units + NWV - 1) / NWV;
    if (wgs > 256 / NGRP) wgs = 256 / NGRP;
    else if (a.xcd_local && a.nimg % 8 == 0 && wgs % 8 != 0 && wgs + 8 - wgs % 8 <= 256 / NGRP) wgs += 8 - wgs % 8;
    auto kern = wino_kernel<C0, C1, COUT, H, W, POOL, RELU>;
    if (lds > 48 * 1024) giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
    GIGA_LAUNCH(kern, dim3(wgs * NGRP), dim3(NWV * 64), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
