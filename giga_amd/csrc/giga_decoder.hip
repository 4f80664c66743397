// Fused implicit decoder: plane-feature gather + all requested LocalDecoder heads in ONE kernel.
//
// Replaces, per head, reference LocalDecoder.forward (ConvONets/conv_onet/models/decoder.py:133-176):
//   3x F.grid_sample (decoder.py:117-122) -> cat -> fc_p -> 5x(fc_c + ResnetBlockFC) -> fc_out
// and the head epilogues of ConvolutionalOccupancyNetwork.decode (models/__init__.py:111-124).
// The reference re-gathers the 96-d feature for every head; here it is gathered once per point,
// kept in registers as MFMA B-operands and reused by every head.
//
// Structure (CDNA4): one wave owns T tiles of 32 query points.  Every layer is computed TRANSPOSED,
// D[feature][point] = W[feature][k] * X[k][point]: weights are the MFMA A operand (streamed from an
// LDS image of the head's fragment blob), activations the B operand.  In that orientation the D
// registers of a lane (16 features of ONE point) are, after ReLU (+ f16 rounding), exactly the B
// operand of the next layer for a permuted contraction order that is baked into the packed weights,
// so the whole 11-layer chain runs without any cross-lane traffic.  The fp32 residual stream lives
// in the accumulator (C-in = stream), biases ride in a constant-one k-slot or in the C operand.
#include <cstdlib>
#include <type_traits>

#include "giga_dev.h"
#include "giga_args.h"

namespace giga {


__device__ __forceinline__ void store_head(const DecArgs& a, int h, long long g, float d0, float d1,
                                           float d2, float d3) {
    const int id = a.head_id[h];
    float* o = a.out[h];
    if (id == 1) {
        if (a.post) {   // F.normalize(dim=2): x / max(||x||_2, 1e-12)
            float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
            float inv = 1.0f / fmaxf(nrm, 1e-12f);
            d0 *= inv; d1 *= inv; d2 *= inv; d3 *= inv;
        }
        *reinterpret_cast<float4*>(o + 4 * g) = make_float4(d0, d1, d2, d3);
    } else {
        if (id == 0 && a.post) d0 = 1.0f / (1.0f + expf(-d0));
        o[g] = d0;
    }
}


// =============================== f16 MFMA path =====================================================
// Persistent workgroups of 8 waves (2 per SIMD).  Round = 512 points (T=2 tiles of 32 per wave): the
// gathered features stay in registers for all heads.  Head weight images (59 KiB) are DOUBLE-BUFFERED
// in LDS and filled by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip): while the MFMA chain of
// step s runs out of buffer s&1, the image of step s+1 streams into the other buffer.  One workgroup
// barrier per step; it is both "image s has landed" and "everyone left image s-1".
constexpr int DEC16_CHUNKS = (int)(DEC16_BYTES / FRAG);      // 59

template <int NW, int CHUNKS = DEC16_CHUNKS>
__device__ __forceinline__ void dma_head_image(const uint8_t* src, uint8_t* lds_dst, int wave, int lane) {
    for (int c = wave; c < CHUNKS; c += NW)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (size_t)c * FRAG + lane * 16),
            (__attribute__((address_space(3))) void*)(lds_dst + c * FRAG), 16, 0, 0);
}

template <int T, bool LATTICE, int NW>
__global__ __launch_bounds__(NW * 64) void decoder_f16_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const half_t* planes = reinterpret_cast<const half_t*>(a.planes);
    const size_t plane_stride = LATTICE ? (size_t)a.B * a.R * a.R * CD : (size_t)a.B * RES * RES * CD;
    // Every workgroup owns one contiguous, equally sized range of 32-point tiles and walks it in rounds of NW*T
    // tiles, so the ragged last round is a PARTIAL round on every CU (fewer active waves per SIMD) instead of a
    // full extra round on some CUs while the others idle.
    // (workgroup i runs on XCD i % 8: the ranges are handed out so that an XCD owns a contiguous eighth of the points, i.e. an
    //  eighth of the scenes, whose planes then stay in its own L2 -- 171 MB of fabric traffic per 32-scene launch otherwise)
    const long long tiles_total = (a.P + 31) / 32;
    const int bid = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const long long tile_lo = tiles_total * bid / gridDim.x;
    const long long tile_hi = tiles_total * (bid + 1) / gridDim.x;
    const int rounds = (int)((tile_hi - tile_lo + NW * T - 1) / (NW * T));
    if (rounds <= 0) return;
    // Step s uses the weight image of head s % nheads.  Waves 4..7 (the second wave of every SIMD) run
    // one step behind waves 0..3 and visit the heads in rotated order (1,2,..,0): while one wave of a
    // SIMD gathers features for its next 64 points, the other is inside its MFMA chain.
    const int phase = a.nheads > 1 ? (wave >> 2) % a.nheads : 0;
    const int work_steps = rounds * a.nheads;
    const int max_phase = a.nheads > 1 ? ((NW / 4 - 1) < (a.nheads - 1) ? (NW / 4 - 1) : (a.nheads - 1)) : 0;
    const int nsteps = work_steps + max_phase;

    dma_head_image<NW>(a.blob + a.head_off[0], smem, wave, lane);

    half8 cf[T][6], ax[T];
    long long gidx[T];
    bool valid[T];
    for (int s = 0; s < nsteps; ++s) {
        const int h = s % a.nheads;                           // head whose image is in buffer s&1
        uint8_t* buf = smem + (a.nheads > 1 ? (s & 1) : 0) * DEC16_BYTES;
        if (a.nheads > 1 || s == 0) {
            // vmcnt(0): my share of image s has landed.  The BUILTIN form (not inline asm) so that hipcc's
            // waitcnt scoreboard knows no LDS-DMA is pending afterwards; otherwise it drains vmcnt(0) at
            // every use of an ordinary global load in the gather below (simm16: vm=0, exp=7, lgkm=15).
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();                                  // everyone's share; everyone left image s-1
        }
        // The LDS-DMA of the next image is issued AFTER this step's gather: with a DMA in flight hipcc
        // waits vmcnt(0) at every use of an ordinary global load, which would serialise the 48 gather
        // loads of a wave behind the 59 KiB transfer.  The MFMA chain below gives the DMA ~2 us to land.
        const bool dma_next = a.nheads > 1 && s + 1 < nsteps;
        const uint8_t* dma_src = a.blob + a.head_off[(s + 1) % a.nheads];
        uint8_t* dma_dst = smem + ((s + 1) & 1) * DEC16_BYTES;
        const int ls = s - phase;                             // this wave's own step counter
        // first tile of this wave in its current round; a wave without tiles (staggered edge step, or the
        // ragged last round) only keeps the image pipeline going
        const long long tile0 = tile_lo + ((long long)(ls / a.nheads) * NW + wave) * T;
        if (ls < 0 || ls >= work_steps || tile0 >= tile_hi) {
            if (dma_next) dma_head_image<NW>(dma_src, dma_dst, wave, lane);
            continue;
        }
        const bool new_round = ls % a.nheads == 0;
        if (LATTICE && new_round) {
            // ---------------- lattice gather: the planes were resampled at the R lattice coordinates
            // (lattice_resample_kernel), so the 96 features of lattice point (ix,iy,iz) are three
            // pixels read straight into B-operand registers: 6 x 16 B per lane, no interpolation.
            const int R = a.R, R2 = R * R;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                long long g = (tile0 + t) * 32 + n;
                valid[t] = tile0 + t < tile_hi && g < a.P;
                if (!valid[t]) g = a.P - 1;
                gidx[t] = g;
                int b, r;
                split_scene(g, a.N, a.invN, b, r);
                const int ix = div_magic(r, a.mR2), rz = r - ix * R2;
                const int iy = div_magic(rz, a.mR), iz = rz - iy * R;
                const float px = a.lin[ix], py = a.lin[iy], pz = a.lin[iz];
                half_t xh = (half_t)px, yh = (half_t)py, zh = (half_t)pz;
                half8 av = {xh, yh, zh, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
                if (hi == 0) {
                    av[3] = (half_t)1.0f;
                    av[4] = (half_t)(px - (float)xh); av[5] = (half_t)(py - (float)yh);
                    av[6] = (half_t)(pz - (float)zh); av[7] = (half_t)1.0f;
                }
                ax[t] = av;
                const half_t* base = planes + (size_t)b * R2 * CD + 8 * hi;
                const int off[3] = {iz * R + ix, iy * R + ix, iz * R + iy};   // (H,W): xz->(z,x) xy->(y,x) yz->(z,y)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        cf[t][2 * pl + hf] = *reinterpret_cast<const half8*>(base + pl * plane_stride +
                                                                             (size_t)off[pl] * CD + 16 * hf);
            }
        }
        if (!LATTICE && new_round) {
            // ---------------- gather: 96 features -> 6 B-operand chunks per tile ------------------
            float pxs[T], pys[T], pzs[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {                    // all coordinate loads first: one round trip
                long long g = (tile0 + t) * 32 + n;
                valid[t] = tile0 + t < tile_hi && g < a.P;
                if (!valid[t]) g = a.P - 1;
                gidx[t] = g;
                pxs[t] = a.p[3 * g + 0]; pys[t] = a.p[3 * g + 1]; pzs[t] = a.p[3 * g + 2];
            }
            // 12 groups (tile, plane, channel half) of 4 tap loads.  Explicit software pipeline: groups
            // 0..5 are issued up front (24 x 16 B loads in flight per lane), then group k is interpolated
            // while group k+6 is issued; the sched_barriers pin that order (left alone, hipcc issues the
            // second tile's taps four at a time with a full round trip each).
            Bilin bl[T][3];
            const half_t* pbase[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float px = pxs[t], py = pys[t], pz = pzs[t];
                int b, rdummy;
                split_scene(gidx[t], a.N, a.invN, b, rdummy);
                const float nx = norm_coord(px), ny = norm_coord(py), nz = norm_coord(pz);
                // aux chunk: [p_hi(3), 1, p_lo(3), 1] on hi=0 lanes, [p_hi(3), 0...] on hi=1 lanes
                half_t xh = (half_t)px, yh = (half_t)py, zh = (half_t)pz;
                half8 av = {xh, yh, zh, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
                if (hi == 0) {
                    av[3] = (half_t)1.0f;
                    av[4] = (half_t)(px - (float)xh); av[5] = (half_t)(py - (float)yh);
                    av[6] = (half_t)(pz - (float)zh); av[7] = (half_t)1.0f;
                }
                ax[t] = av;
                // xz: (u,v)=(x,z)  xy: (x,y)  yz: (y,z)      common.py:246-251
                bl[t][0] = bilin_setup(nx, nz);
                bl[t][1] = bilin_setup(nx, ny);
                bl[t][2] = bilin_setup(ny, nz);
                pbase[t] = planes + (size_t)b * RES * RES * CD + 8 * hi;
            }
            constexpr int NG = T * 6, DEPTH = 6;
            half8 raw[NG][4];
            auto issue = [&](int k) {
                const int t = k / 6, pl = (k % 6) / 2, hf = k % 2;
                const half_t* base = pbase[t] + pl * plane_stride + 16 * hf;
                raw[k][0] = *reinterpret_cast<const half8*>(base + (size_t)bl[t][pl].o00 * CD);
                raw[k][1] = *reinterpret_cast<const half8*>(base + (size_t)bl[t][pl].o01 * CD);
                raw[k][2] = *reinterpret_cast<const half8*>(base + (size_t)bl[t][pl].o10 * CD);
                raw[k][3] = *reinterpret_cast<const half8*>(base + (size_t)bl[t][pl].o11 * CD);
            };
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) issue(k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int t = k / 6, pl = (k % 6) / 2;
                half8 r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float acc = (float)raw[k][0][j] * bl[t][pl].w00;
                    acc = fmaf((float)raw[k][1][j], bl[t][pl].w01, acc);
                    acc = fmaf((float)raw[k][2][j], bl[t][pl].w10, acc);
                    acc = fmaf((float)raw[k][3][j], bl[t][pl].w11, acc);
                    r[j] = (half_t)acc;
                }
                cf[t][k % 6] = r;
                if (k + DEPTH < NG) issue(k + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (dma_next) dma_head_image<NW>(dma_src, dma_dst, wave, lane);
        // ---------------- MFMA chain of head h out of buffer s&1 --------------------------------------
        {
            const half8* W = reinterpret_cast<const half8*>(buf);
            const float* ctab = reinterpret_cast<const float*>(buf + (size_t)DEC16_FRAGS * FRAG);
            // Fragment indices in the image: block b -> fc_c 11b..11b+6 (6 feature chunks + aux), fc_0 11b+7,8,
            // fc_1 11b+9,10; tail: aux (fc_1 bias of the last block) 55, fc_out 56,57.
            // Every term added to the residual stream is an accumulation into `net`, so their order is
            // free: the 7 fc_c MFMAs of block b+1 do not depend on block b's MLP and are issued BETWEEN
            // fc_0 and fc_1 of block b, where they fill the MFMA -> VALU (relu/cvt) -> MFMA latency bubble.
            f32x16 net[T], hh[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) net[t][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const half8 A = W[c * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) net[t] = mfma16(A, c < 6 ? cf[t][c] : ax[t], net[t]);
            }
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const int k = 11 * blk;
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + blk * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
                {   // hh = fc_0(relu(net)) + b0 : the stream is read (packed) here ...
                    const half8 A0 = W[(k + 7) * 64 + lane], A1 = W[(k + 8) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        hh[t] = mfma16(A0, pack_relu8(net[t], 0), c0);
                        hh[t] = mfma16(A1, pack_relu8(net[t], 1), hh[t]);
                    }
                }
                if (blk + 1 < NBLK) {   // ... so the next block's fc_c (+ folded biases) may already accumulate
#pragma unroll
                    for (int c = 0; c < 7; ++c) {
                        const half8 A = W[(k + 11 + c) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < T; ++t) net[t] = mfma16(A, c < 6 ? cf[t][c] : ax[t], net[t]);
                    }
                } else {                // last block: only its fc_1 bias remains (aux fragment 55)
                    const half8 A = W[55 * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) net[t] = mfma16(A, ax[t], net[t]);
                }
                {   // net += fc_1(relu(hh))
                    const half8 A0 = W[(k + 9) * 64 + lane], A1 = W[(k + 10) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        net[t] = mfma16(A0, pack_relu8(hh[t], 0), net[t]);
                        net[t] = mfma16(A1, pack_relu8(hh[t], 1), net[t]);
                    }
                }
            }
            {   // fc_out(relu(net))
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + NBLK * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
                const half8 A0 = W[56 * 64 + lane], A1 = W[57 * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    f32x16 o = mfma16(A0, pack_relu8(net[t], 0), c0);
                    o = mfma16(A1, pack_relu8(net[t], 1), o);
                    if (hi == 0 && valid[t]) store_head(a, h, gidx[t], o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}


// =============================== f16x3 split-operand MFMA path =====================================
// Same transposed, lane-local chain on v_mfma_f32_32x32x16_f16, but every operand is a PAIR hi = f16(v), lo = f16(v - hi)
// and every product is evaluated as  W_lo*x_hi + W_hi*x_lo + W_hi*x_hi  with fp32 accumulation (the dropped W_lo*x_lo term is
// 2^-22 relative): operands are carried to ~22 bits at 3x the f16 MFMA count, i.e. fp32-grade results (<= 1e-5 on the head
// outputs against the fp32 oracle) at ~5x the rate of the fp32-input MFMA.  This is the mode that satisfies both clauses of the
// north star: "MFMA fp16 tiles" and "within 1e-3".
// Structure: HEAD-RESIDENT workgroups.  A workgroup serves ONE head for its whole life: the 111 KiB [hi, lo] image of that
// head is filled into LDS once (LDS-DMA) and there is no barrier, no weight traffic and no head switching afterwards;
// workgroup i takes head i % nheads and an equal contiguous share of the 32-point tiles.  The features are gathered per
// head (three pixels of 128 B per point on the lattice path -- the planes were resampled AND split by
// lattice_resample_split_kernel, so the B operands are loaded ready-made; 12 bilinear taps on fp32 planes otherwise), which
// costs a few percent of the 162-MFMA chain.  8 waves of T = 2 tiles.
constexpr int DEC16S_CHUNKS = (int)(DEC16S_BYTES / FRAG);    // 111

// SPLIT = false runs the same head-resident structure with single f16 operands (the plain f16 image of 59 KiB, f16 planes):
// it serves small launches of the fp16 mode, where decoder_f16_kernel's shared-feature rounds leave most CUs idle.
#ifdef GIGA_TRACE   // diagnostic build: issue timeline (s_memtime) of one workgroup of the head-resident decoder
static __device__ long long g_dec_trace[16 * 16];
#define DEC_T(idx) do { if (blockIdx.x == 100 && lane == 0 && (idx) < 16) g_dec_trace[wave * 16 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEC_T(idx) do {} while (0)
#endif
template <int T, bool LATTICE, int NW, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void decoder_f16s_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NCH = SPLIT ? DEC16S_CHUNKS : DEC16_CHUNKS;
    constexpr int NFR = SPLIT ? DEC16S_FRAGS : DEC16_FRAGS;
    constexpr int BLK = SPLIT ? DEC16S_BLK : 11;               // fragments per block
    constexpr int PR = SPLIT ? 2 : 1;                          // fragments per weight chunk ([hi, lo] pair or single)
    constexpr int F_AUX = 6 * PR, F_FC0 = F_AUX + 1, F_FC1 = F_FC0 + 2 * PR, F_TAIL = NBLK * BLK, F_OUT = F_TAIL + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    // workgroup -> (head, slot of the tile range).  Workgroup i runs on XCD i % 8; when the grid is a multiple of 8 * nheads the
    // slots of one XCD are made contiguous (an eighth of the points = an eighth of the scenes: their planes stay in that XCD's
    // L2; with the plain i % nheads / i / nheads map every XCD walked every scene: 527 MB of fabric traffic per launch)
    const int slots = gridDim.x / a.nheads;
    int hsel = blockIdx.x % a.nheads, slot = blockIdx.x / a.nheads;
    if (gridDim.x % (8 * a.nheads) == 0) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, spx = slots >> 3;
        hsel = k % a.nheads;
        slot = xcd * spx + k / a.nheads;
    }
    DEC_T(0);
    dma_head_image<NW, NCH>(a.blob + a.head_off[hsel], smem, wave, lane);
    DEC_T(1);

    const long long tiles_total = (a.P + 31) / 32;
    const long long tile_lo = tiles_total * slot / slots;
    const long long tile_hi = tiles_total * (slot + 1) / slots;
    const half8* W = reinterpret_cast<const half8*>(smem);
    const float* ctab = reinterpret_cast<const float*>(smem + (size_t)NFR * FRAG);
    const size_t plane_stride = LATTICE ? (size_t)a.B * a.R * a.R * CD : (size_t)a.B * RES * RES * CD;   // in features

    half8 cfh[T][6], cfl[T][6], ax[T];
    long long gidx[T];
    bool valid[T];
    long long tile0 = tile_lo + (long long)wave * T;
    for (int iter = 0;; ++iter, tile0 += NW * T) {
        const bool has = tile0 < tile_hi;
        if (has) {
            float pxs[T], pys[T], pzs[T];
            int bs[T], rs[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                long long g = (tile0 + t) * 32 + n;
                valid[t] = tile0 + t < tile_hi && g < a.P;
                if (!valid[t]) g = a.P - 1;
                gidx[t] = g;
                split_scene(g, a.N, a.invN, bs[t], rs[t]);
                if constexpr (LATTICE) {
                    const int R = a.R, R2 = R * R;
                    const int ix = div_magic(rs[t], a.mR2), rz = rs[t] - ix * R2;
                    const int iy = div_magic(rz, a.mR), iz = rz - iy * R;
                    pxs[t] = a.lin[ix]; pys[t] = a.lin[iy]; pzs[t] = a.lin[iz];
                    const int off[3] = {iz * R + ix, iy * R + ix, iz * R + iy};   // (H,W): xz->(z,x) xy->(y,x) yz->(z,y)
                    if constexpr (SPLIT) {
                        // split planes: pixel = 4 groups of 8 channels x [8 hi halfs | 8 lo halfs]; chunk c = 2*pl + hf of lane
                        // half `hi` is group 2*hf + hi of plane pl  ->  one 32-byte read per chunk
                        const half_t* base = reinterpret_cast<const half_t*>(a.planes) + 2 * ((size_t)bs[t] * R2 * CD) + 16 * hi;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                            for (int hf = 0; hf < 2; ++hf) {
                                const half_t* q = base + 2 * (pl * plane_stride + (size_t)off[pl] * CD) + 32 * hf;
                                cfh[t][2 * pl + hf] = *reinterpret_cast<const half8*>(q);
                                cfl[t][2 * pl + hf] = *reinterpret_cast<const half8*>(q + 8);
                            }
                    } else {
                        const half_t* base = reinterpret_cast<const half_t*>(a.planes) + (size_t)bs[t] * R2 * CD + 8 * hi;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                            for (int hf = 0; hf < 2; ++hf)
                                cfh[t][2 * pl + hf] = *reinterpret_cast<const half8*>(base + pl * plane_stride +
                                                                                      (size_t)off[pl] * CD + 16 * hf);
                    }
                } else {
                    pxs[t] = a.p[3 * g + 0]; pys[t] = a.p[3 * g + 1]; pzs[t] = a.p[3 * g + 2];
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                // aux chunk: [p_hi(3), 1, p_lo(3), 1] on hi=0 lanes, [p_hi(3), 0...] on hi=1 lanes (exact by itself)
                const float px = pxs[t], py = pys[t], pz = pzs[t];
                half_t xh = (half_t)px, yh = (half_t)py, zh = (half_t)pz;
                half8 av = {xh, yh, zh, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
                if (hi == 0) {
                    av[3] = (half_t)1.0f;
                    av[4] = (half_t)(px - (float)xh); av[5] = (half_t)(py - (float)yh);
                    av[6] = (half_t)(pz - (float)zh); av[7] = (half_t)1.0f;
                }
                ax[t] = av;
            }
            if constexpr (!LATTICE) {
                // ---------------- generic gather: per (tile, plane, channel half) the lane's 8 channels of the 4 bilinear
                // taps, interpolated in fp32 in aten's tap order, then split (or rounded).  Software pipeline of DEPTH groups.
                typedef typename std::conditional<SPLIT, float, half_t>::type PT;
                const PT* planes = reinterpret_cast<const PT*>(a.planes);
                Bilin bl[T][3];
                const PT* pbase[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float nx = norm_coord(pxs[t]), ny = norm_coord(pys[t]), nz = norm_coord(pzs[t]);
                    bl[t][0] = bilin_setup(nx, nz);      // xz: (u,v)=(x,z)  xy: (x,y)  yz: (y,z)      common.py:246-251
                    bl[t][1] = bilin_setup(nx, ny);
                    bl[t][2] = bilin_setup(ny, nz);
                    pbase[t] = planes + (size_t)bs[t] * RES * RES * CD + 8 * hi;
                }
                constexpr int NG = T * 6, DEPTH = SPLIT ? 3 : 6;
                constexpr int VPT = SPLIT ? 2 : 1;            // 16-byte vectors per tap
                uint4 raw[NG][4][VPT];
                auto issue = [&](int k) {
                    const int t = k / 6, pl = (k % 6) / 2, hf = k % 2;
                    const PT* base = pbase[t] + pl * plane_stride + 16 * hf;
                    const int o[4] = {bl[t][pl].o00, bl[t][pl].o01, bl[t][pl].o10, bl[t][pl].o11};
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                        for (int q = 0; q < VPT; ++q)
                            raw[k][tp][q] = *reinterpret_cast<const uint4*>(base + (size_t)o[tp] * CD + q * (16 / sizeof(PT)));
                };
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) issue(k);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int t = k / 6, pl = (k % 6) / 2;
                    const Bilin& B4 = bl[t][pl];
                    float v[8];
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const f32x4 v00 = __builtin_bit_cast(f32x4, raw[k][0][q]), v01 = __builtin_bit_cast(f32x4, raw[k][1][q]);
                            const f32x4 v10 = __builtin_bit_cast(f32x4, raw[k][2][q]), v11 = __builtin_bit_cast(f32x4, raw[k][3][q]);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v[4 * q + e] = fmaf(v11[e], B4.w11, fmaf(v10[e], B4.w10, fmaf(v01[e], B4.w01, v00[e] * B4.w00)));
                        }
                        split8(v, cfh[t][k % 6], cfl[t][k % 6]);
                    } else {
                        const half8 h00 = __builtin_bit_cast(half8, raw[k][0][0]), h01 = __builtin_bit_cast(half8, raw[k][1][0]);
                        const half8 h10 = __builtin_bit_cast(half8, raw[k][2][0]), h11 = __builtin_bit_cast(half8, raw[k][3][0]);
                        half8 r;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float acc = (float)h00[j] * B4.w00;
                            acc = fmaf((float)h01[j], B4.w01, acc);
                            acc = fmaf((float)h10[j], B4.w10, acc);
                            acc = fmaf((float)h11[j], B4.w11, acc);
                            r[j] = (half_t)acc;
                        }
                        cfh[t][k % 6] = r;
                    }
                    if (k + DEPTH < NG) issue(k + DEPTH);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        DEC_T(2 + 3 * iter);
        if (iter == 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): my share of the weight image has landed
            __syncthreads();                                  // everyone's share (all waves run iteration 0)
        }
        if (!has) break;
        DEC_T(3 + 3 * iter);
        // ---------------- the chain.  Fragment indices per block b (PR = fragments per chunk): fc_c BLK*b + PR*c, aux
        // BLK*b + 6*PR, fc_0 .. + 1 + PR*c, fc_1 .. + 1 + 2*PR + PR*c; tail aux NBLK*BLK, fc_out + 1 + PR*c
        f32x16 net[T], hh[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) net[t][r] = 0.f;
        // acc[t] += W(frag) * X[t]  for one k-chunk: three MFMAs on [hi, lo] pairs, one on single operands
        auto mm = [&](int frag, const half8 (&xh)[T], const half8 (&xl)[T], f32x16 (&acc)[T]) {
            const half8 Ah = W[frag * 64 + lane];
            if constexpr (SPLIT) {
                const half8 Al = W[(frag + 1) * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[t] = mfma16(Al, xh[t], acc[t]);
                    acc[t] = mfma16(Ah, xl[t], acc[t]);
                    acc[t] = mfma16(Ah, xh[t], acc[t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = mfma16(Ah, xh[t], acc[t]);
            }
        };
        auto fc_c = [&](int blk) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                half8 xh[T], xl[T];
#pragma unroll
                for (int t = 0; t < T; ++t) { xh[t] = cfh[t][c]; xl[t] = cfl[t][c]; }
                mm(BLK * blk + PR * c, xh, xl, net);
            }
            const half8 A = W[(BLK * blk + F_AUX) * 64 + lane];
#pragma unroll
            for (int t = 0; t < T; ++t) net[t] = mfma16(A, ax[t], net[t]);
        };
        auto ctab_regs = [&](int row) {
            f32x16 c0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ctab + row * 32 + 8 * q + 4 * hi);
                c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
            }
            return c0;
        };
        // dst[t] += W(frag0 + PR*c) * relu(src[t])
        auto dense = [&](int frag0, const f32x16 (&src)[T], f32x16 (&dst)[T]) {
            half8 xh[2][T], xl[2][T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if constexpr (SPLIT) split_relu8(src[t], c, xh[c][t], xl[c][t]);
                    else xh[c][t] = pack_relu8(src[t], c);
                }
#pragma unroll
            for (int c = 0; c < 2; ++c) mm(frag0 + PR * c, xh[c], xl[c], dst);
        };
        fc_c(0);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const f32x16 c0 = ctab_regs(blk);
#pragma unroll
            for (int t = 0; t < T; ++t) hh[t] = c0;
            dense(BLK * blk + F_FC0, net, hh);                // hh = fc_0(relu(net)) + b0
            // every term of the residual stream is an accumulation, so the next block's fc_c (+ folded biases) is issued
            // here, where it covers the conversion of hh
            if (blk + 1 < NBLK) fc_c(blk + 1);
            else {
                const half8 A = W[F_TAIL * 64 + lane];       // tail aux: fc_1 bias of the last block
#pragma unroll
                for (int t = 0; t < T; ++t) net[t] = mfma16(A, ax[t], net[t]);
            }
            dense(BLK * blk + F_FC1, hh, net);                // net += fc_1(relu(hh))
        }
        {
            const f32x16 c0 = ctab_regs(NBLK);
            f32x16 o[T];
#pragma unroll
            for (int t = 0; t < T; ++t) o[t] = c0;
            dense(F_OUT, net, o);                             // fc_out(relu(net))
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (hi == 0 && valid[t]) store_head(a, hsel, gidx[t], o[t][0], o[t][1], o[t][2], o[t][3]);
        }
        DEC_T(4 + 3 * iter);
    }
    DEC_T(15);
}


// =============================== lattice decoder with separable fc_c ("G lines") ====================
// The inference lattice (detection_implicit.py:28-31) is a product grid, and the three planes each see only two of its three
// coordinates: point (ix, iy, iz) reads pixel (iz, ix) of plane xz, (iy, ix) of plane xy and (iz, iy) of plane yz.  fc_c is
// LINEAR in the concatenated feature (decoder.py:169) and so is fc_p in p (decoder.py:165), hence for every block b
//     fc_c[b](c) + bias  =  G_xz[b][iz]  +  G_xy[b][iy]  +  Wc_b[:, 64:96] f_yz(iz, iy)          at fixed (scene, ix)
//     G_xz[b][iz] = Wc_b[:, 0:32] f_xz(iz, ix) (+ the z term of fc_p for b = 0),   G_xy[b][iy] = Wc_b[:, 32:64] f_xy(iy, ix) + biases
//                   (+ the x and y terms of fc_p for b = 0)
// A workgroup takes a SLAB = (scene, ix): R*R points.  LINE PHASE: it evaluates the two "lines" G_xz[.][0..R) and G_xy[.][0..R) of
// all five blocks (22 jobs of 3-7 MFMAs spread over the waves; operands straight from L2) and parks them in LDS as [hi, lo] f16
// pairs (hi + lo carries 22 bits: fp32-grade).  TILE PHASE: a tile is 4 iy x 8 iz points, and a line value enters the residual
// stream through the MATRIX pipe, not the VALU: A operand = the line values of the tile's 4 iy (8 iz) for the lane's feature --
// one ds_read_b128 --, B operand = a constant one-hot selector "is this point's iy (iz) the k-slot's?", so that
//     net += A_xz(blk) * SEL_z ;  net += A_xy(blk) * SEL_y          (two K = 16 MFMAs, no VALU instruction, no per-point data)
// and only the yz part of fc_c is still a product with per-point features: 4 MFMAs per block instead of 7 for fc_c (+ aux), 43
// per tile and head instead of 58 (f16x3: 107 instead of 162), with the xz / xy contributions -- two thirds of fc_c, all of fc_p
// and every bias of the stream -- exact to 2^-22 in every f16-class mode.  Same fragments as the other f16 kernels (the head
// image of giga_pack.cpp; nothing new is packed): the tile phase keeps only what it uses resident in LDS (yz chunks of fc_c, fc_0,
// fc_1, fc_out, the C table: 33 KiB, f16x3 65 KiB), the line phase reads its weight fragments from the image in L2.
// Head-resident persistent workgroups as decoder_f16s_kernel; slabs are handed out so that an XCD owns a contiguous eighth of
// the scenes.  Needs R % 8 == 0 (R = 40: 50 tiles per slab, none ragged).
constexpr int LAT_ROWS = 2 * NBLK + 1;     // line rows: G_xz[0..4], G_xy[0..4], and the bias of the last fc_1 as G_xy[5]
constexpr int LAT_MAX_R = 40;
// the low half of a line value is stored scaled by 2^11 and its selector slot carries 2^-11: hi and lo then have the same
// magnitude class, so lo is a NORMAL f16 number whenever the value itself is above 2^-13 (unscaled it would be subnormal for
// every |value| < 2^-3), and the product lo * 2^-11 is exact
constexpr float LAT_LO_SCALE = 2048.0f, LAT_LO_INV = 1.0f / 2048.0f;
template <bool SPLIT, bool DBUF>
constexpr size_t lat_lds_bytes(int R) {
    return (size_t)(32 * (SPLIT ? 2 : 1) + 1) * FRAG + (size_t)(DBUF ? 2 : 1) * LAT_ROWS * (R / 4) * 32 * 16 + 16;
}

// WORK DISTRIBUTION inside a workgroup is DYNAMIC: tiles (and, with two line buffers, the next slab's line jobs) are items of a
// pool that the waves drain through one LDS counter.  A static tile -> wave map loses a third of the slab: the SIMD arbiter
// favours its older waves (s_memtime trace, profiles/r03: per tile 4.8 k clocks on waves 0-3, 5.5 k on 4-7, 7.6 k on 8-11), so
// the old waves idle at the slab barrier while the young ones still owe two tiles.
// DBUF: two line buffers (fits beside the 33-KiB image of the plain mode, not beside the 65 KiB of f16x3).  Pool s = the tiles of
// slab s + the line jobs of slab s + 1 (they fill the other buffer); ONE workgroup barrier per slab.  Single buffer: pool s = the
// tiles of slab s; barrier, line phase of slab s + 1 (all waves), barrier.
template <bool SPLIT, int NW, bool DBUF>
__global__ __launch_bounds__(NW * 64) void decoder_lat_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int PR = SPLIT ? 2 : 1;                          // fragments per weight chunk ([hi, lo] pair or single)
    constexpr int BLK = SPLIT ? DEC16S_BLK : 11;               // fragments per block in the head image
    constexpr int NFR = SPLIT ? DEC16S_FRAGS : DEC16_FRAGS;
    constexpr int F_AUX = 6 * PR, F_TAIL = NBLK * BLK, F_OUT = F_TAIL + 1;
    constexpr int WF = 32 * PR;                                // resident fragments: per block [yz c0, yz c1, fc_0 x2, fc_1 x2], then fc_out x2
    constexpr int NJOB = 2 * LAT_ROWS;                         // line jobs of a slab: (row, half of the line)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int slots = gridDim.x / a.nheads;
    int hsel = blockIdx.x % a.nheads, slot = blockIdx.x / a.nheads;
    if (gridDim.x % (8 * a.nheads) == 0) {                     // XCD-contiguous slab ranges (workgroup i runs on XCD i % 8)
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, spx = slots >> 3;
        hsel = k % a.nheads;
        slot = xcd * spx + k / a.nheads;
    }
    const int R = a.R, R2 = R * R, RG = R >> 2;                // RG: groups of 4 line pixels
    // work units: a slab (scene, ix), or for small launches one of NP equal tile ranges of it (a.lat_parts divides the tile count);
    // a unit has its own pool and its own copy of the slab's lines
    const int NP = a.lat_parts;
    const int nslab = a.B * R * NP;                            // (units)
    const int slab_lo = (int)((long long)nslab * slot / slots), slab_hi = (int)((long long)nslab * (slot + 1) / slots);
    if (slab_lo >= slab_hi) return;
    DEC_T(0);
    const uint8_t* img = a.blob + a.head_off[hsel];
    for (int d = wave; d <= WF; d += NW) {                     // the resident part of the head image (+ the C table chunk)
        int src;
        if (d < 30 * PR) {
            const int b = d / (6 * PR), e = d - b * 6 * PR;
            src = e < 2 * PR ? BLK * b + 4 * PR + e : BLK * b + F_AUX + 1 + (e - 2 * PR);
        } else if (d < WF) {
            src = F_OUT + (d - 30 * PR);
        } else {
            src = NFR;
        }
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(img + (size_t)src * FRAG + lane * 16),
            (__attribute__((address_space(3))) void*)(smem + d * FRAG), 16, 0, 0);
    }
    const half8* W = reinterpret_cast<const half8*>(smem);
    const float* ctab = reinterpret_cast<const float*>(smem + (size_t)WF * FRAG);
    // lines: [row][group of 4 pixels][feature 0..31][pixel % 4] x (hi, lo * 2^11) halfs = 16 bytes per (row, group, feature)
    uint8_t* G0 = smem + (size_t)(WF + 1) * FRAG;
    const size_t gbytes = (size_t)LAT_ROWS * RG * 32 * 16;
    unsigned* counter = reinterpret_cast<unsigned*>(G0 + (DBUF ? 2 : 1) * gbytes);
    const half8* Wg = reinterpret_cast<const half8*>(img);
    const half_t* planes = reinterpret_cast<const half_t*>(a.planes);
    const size_t plane_stride = (size_t)a.B * R2 * CD;         // features per plane
    const int tz_n = R >> 3, ntile = (R >> 2) * tz_n;          // tiles of 4 iy x 8 iz
    const int tcount = ntile / NP;                             // tiles per unit
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // one-hot selectors (B operands; constant): this lane's point sits at (dy, dz) = (n >> 3, n & 7) of its tile.  k-slot pair
    // j of SEL_y (hi = 0 lanes only) is iy0 + j; k-slot pair j of SEL_z is iz0 + 4 hi + j.  The halves of a pair carry
    // (1, 2^-11): the A operand holds (hi, 2^11 lo) of the line value there.
    half8 sel_y = zero8, sel_z = zero8;
    {
        const int dy = n >> 3, dz = n & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool my = hi == 0 && dy == j, mz = dz == 4 * hi + j;
            sel_y[2 * j] = my ? (half_t)1.0f : (half_t)0.0f; sel_y[2 * j + 1] = my ? (half_t)LAT_LO_INV : (half_t)0.0f;
            sel_z[2 * j] = mz ? (half_t)1.0f : (half_t)0.0f; sel_z[2 * j + 1] = mz ? (half_t)LAT_LO_INV : (half_t)0.0f;
        }
    }

    // features of pixel `off` of plane `pl`, scene b: chunk hf (16 channels; this lane's 8) as B operands
    auto load_feat = [&](int b, int pl, int off, int hf, half8& fh, half8& fl) {
        if constexpr (SPLIT) {
            const half_t* q = planes + 2 * ((size_t)b * R2 * CD + pl * plane_stride + (size_t)off * CD) + 16 * hi + 32 * hf;
            fh = *reinterpret_cast<const half8*>(q);
            fl = *reinterpret_cast<const half8*>(q + 8);
        } else {
            fh = *reinterpret_cast<const half8*>(planes + (size_t)b * R2 * CD + pl * plane_stride + (size_t)off * CD + 16 * hf + 8 * hi);
            fl = zero8;
        }
    };
    // acc += W(frag) * x : three MFMAs on [hi, lo] pairs, one on single operands
    auto mm1 = [&](const half8* Wsrc, int frag, const half8& xh, const half8& xl, f32x16& acc) {
        const half8 Ah = Wsrc[frag * 64 + lane];
        if constexpr (SPLIT) {
            const half8 Al = Wsrc[(frag + 1) * 64 + lane];
            acc = mfma16(Al, xh, acc);
            acc = mfma16(Ah, xl, acc);
            acc = mfma16(Ah, xh, acc);
        } else {
            acc = mfma16(Ah, xh, acc);
        }
    };
    auto ctab_regs = [&](int row) {
        f32x16 c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(ctab + row * 32 + 8 * q + 4 * hi);
            c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
        }
        return c0;
    };
    // dst += W(frag0 + PR*c) * relu(src), c = 0, 1 (resident fragments)
    auto dense = [&](int frag0, const f32x16& src, f32x16& dst) {
        half8 xh[2], xl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (SPLIT) split_relu8(src, c, xh[c], xl[c]);
            else { xh[c] = pack_relu8(src, c); xl[c] = zero8; }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) mm1(W, frag0 + PR * c, xh[c], xl[c], dst);
    };
    // ---- one line job of slab `slab`: row (job >> 1), pixels 32 (job & 1) .. + 31, into line buffer G
    auto line_job = [&](int unit, int job, uint8_t* G) {
        const int slab = unit / NP;
        const int b = slab / R, ix = slab - b * R;             // (uniform)
        const float px_ = a.lin[ix];
        const int row = job >> 1, ct = job & 1;
        if (32 * ct >= R) return;
        const bool xy = row >= NBLK;
        const int blk = xy ? row - NBLK : row;
        int px = 32 * ct + n;
        const bool pvalid = px < R;
        px = pvalid ? px : R - 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (blk < NBLK) {
            const int pl = xy ? 1 : 0;                          // plane xz: pixel (H = iz, W = ix); plane xy: (H = iy, W = ix)
            half8 fh[2], fl[2];
            load_feat(b, pl, px * R + ix, 0, fh[0], fl[0]);
            load_feat(b, pl, px * R + ix, 1, fh[1], fl[1]);
#pragma unroll
            for (int c = 0; c < 2; ++c) mm1(Wg, BLK * blk + PR * (2 * pl + c), fh[c], fl[c], acc);
        }
        if (xy || blk == 0) {
            // aux fragment: k-slots [p_hi(3), 1, p_lo(3), 1] on hi = 0 lanes, [p_hi(3), 0...] on hi = 1 lanes.  fc_p is linear in p:
            // the xy line carries the x and y terms and the constant-one slots (the biases), the xz line the z term.
            const float pc = a.lin[px];
            const half_t ch = (half_t)pc, xh_ = (half_t)px_;
            half8 av = zero8;
            if (xy) {
                av[0] = xh_; av[1] = ch;
                if (hi == 0) { av[3] = (half_t)1.0f; av[4] = (half_t)(px_ - (float)xh_); av[5] = (half_t)(pc - (float)ch); av[7] = (half_t)1.0f; }
            } else {
                av[2] = ch;
                if (hi == 0) av[6] = (half_t)(pc - (float)ch);
            }
            acc = mfma16(Wg[(blk < NBLK ? BLK * blk + F_AUX : F_TAIL) * 64 + lane], av, acc);
        }
        if (pvalid) {                                           // this lane: pixel px, features 8q + 4hi + j  ->  (hi, 2^11 lo) pairs
            uint8_t* g = G + ((size_t)(row * RG + (px >> 2)) * 32) * 16 + (px & 3) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const half_t h = (half_t)acc[r];
                const half_t l = (half_t)((acc[r] - (float)h) * LAT_LO_SCALE);
                const half2v pr = {h, l};
                *reinterpret_cast<half2v*>(g + (8 * (r >> 2) + 4 * hi + (r & 3)) * 16) = pr;
            }
        }
    };
    // yz features of tile t of slab `slab_` (this lane's point): requested one item AHEAD -- the L2 / HBM round trip of a tile's
    // features (a few thousand clocks) otherwise stalls every chain at its third MFMA
    auto tile_feat = [&](int unit_, int tl_, half8 (&fh)[2], half8 (&fl)[2]) {
        const int slab_ = unit_ / NP, t_ = (unit_ - slab_ * NP) * tcount + tl_;
        const int b_ = slab_ / R;                              // (uniform)
        const int ty = t_ / tz_n, tz = t_ - ty * tz_n;
        const int iy = 4 * ty + (n >> 3), iz = 8 * tz + (n & 7);
        load_feat(b_, 2, iz * R + iy, 0, fh[0], fl[0]);         // plane yz: pixel (H = iz, W = iy)
        load_feat(b_, 2, iz * R + iy, 1, fh[1], fl[1]);
    };
    // next item of the workgroup's stream (one LDS atomic per item; the value is wave-uniform)
    auto grab = [&]() -> int {
        unsigned g = 0;
        if (lane == 0) g = atomicAdd(counter, 1u);
        return __builtin_amdgcn_readfirstlane((int)g);
    };

    DEC_T(1);
    if (threadIdx.x == 0) *counter = 0u;
    for (int job = wave; job < NJOB; job += NW) line_job(slab_lo, job, G0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): my share of the resident image has landed
    DEC_T(3);
    __syncthreads();
    DEC_T(4);
    const int npool = slab_hi - slab_lo;
    const int POOL = DBUF ? tcount + NJOB : tcount;            // items per pool: the unit's tiles (+ the next unit's line jobs)
    // position k of a pool -> tile index (>= 0) or line job (-1 - job).  With two buffers every third position (2, 5, 8, ...) is a
    // line job of the next slab until the NJOB jobs are out: a line job is a chain of L2 round trips with a handful of MFMAs, and
    // spread like this it runs under the tiles of the SIMD's other waves instead of 22 of them idling the matrix pipes together at
    // the end of the pool; the pool's last positions are tiles.
    const bool mix = tcount >= 2 * NJOB;                       // (small units: the jobs simply follow the tiles)
    auto pool_tile = [&](int k) -> int {
        if constexpr (!DBUF) return k;
        if (!mix) return k < tcount ? k : -1 - (k - tcount);
        if (k < 3 * NJOB) { const int j = (k + 1) / 3; return (k % 3 == 2) ? -j : k - j; }    // (k % 3 == 2: job j - 1 = k / 3)
        return k - NJOB;
    };
    int cur = 0;                                               // the pool whose barrier this wave has not passed yet
    int item = grab();
    half8 nfh[2], nfl[2];
    {
        const int p0 = item / POOL, t0 = pool_tile(item - p0 * POOL);
        if (p0 < npool && t0 >= 0) tile_feat(slab_lo + p0, t0, nfh, nfl);
    }
    int ntiles_done = 0;
    while (true) {
        const int pool = item / POOL, k = item - pool * POOL;
        // slab boundaries between this wave's previous item and this one (every wave passes each of them exactly once)
        const int upto = pool < npool ? pool : npool - 1;
        while (cur < upto) {
            if (cur < 2) DEC_T(8 + 3 * cur);
            __syncthreads();                                    // pool `cur` is drained: its tiles are done (DBUF: and the next lines complete)
            if constexpr (!DBUF) {
                for (int job = wave; job < NJOB; job += NW) line_job(slab_lo + cur + 1, job, G0);
                __syncthreads();
            }
            if (cur < 2) DEC_T(10 + 3 * cur);
            ++cur;
        }
        if (pool >= npool) break;
        const int slab = slab_lo + pool;
        const int nxt = grab();                                 // the item after this one: its features are requested now
        const int t = pool_tile(k);
        if (t < 0) {                                            // (DBUF) a line job of the next slab, into the other buffer
            if (slab + 1 < slab_hi) line_job(slab + 1, -1 - t, G0 + (size_t)((pool + 1) & 1) * gbytes);
            const int pn = nxt / POOL, tn = pool_tile(nxt - pn * POOL);
            if (pn < npool && tn >= 0) tile_feat(slab_lo + pn, tn, nfh, nfl);
            item = nxt;
            continue;
        }
        const int sl = slab / NP, t_g = (slab - sl * NP) * tcount + t;      // the unit's slab; the tile's index inside the slab
        const int b = sl / R, ix = sl - b * R;                 // (uniform)
        const uint8_t* G = G0 + (DBUF ? (size_t)(pool & 1) * gbytes : 0);
        // net += line row `row`, pixel group `grp` (this lane: feature n, the group's 4 pixels as (hi, lo) pairs) through `sel`
        auto add_line = [&](int row, int grp, const half8& sel, f32x16& net) {
            const half8 A = *reinterpret_cast<const half8*>(G + ((size_t)(row * RG + grp) * 32 + n) * 16);
            net = mfma16(A, sel, net);
        };
        {
            // ---------------- one tile: 4 iy x 8 iz points of the slab
            // (compiler barrier: the weight fragments are loop-invariant LDS reads, and hoisting all of them out of this loop --
            //  130-260 registers -- spills the whole chain)
            asm volatile("" ::: "memory");
            const int ty = t_g / tz_n, tz = t_g - ty * tz_n;    // (uniform)
            const int iy = 4 * ty + (n >> 3), iz = 8 * tz + (n & 7);
            const long long gidx = (long long)b * a.N + (long long)ix * R2 + iy * R + iz;
            const int gy = ty, gz = 2 * tz + hi;                // line groups: SEL_y's four iy; this lane half's four iz
            half8 cfh[2], cfl[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { cfh[c] = nfh[c]; cfl[c] = nfl[c]; }
            {
                const int pn = nxt / POOL, tn = pool_tile(nxt - pn * POOL);
                if (pn < npool && tn >= 0) tile_feat(slab_lo + pn, tn, nfh, nfl);
            }
            f32x16 net, hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) net[r] = 0.f;
            add_line(0, gz, sel_z, net);
            add_line(NBLK, gy, sel_y, net);
#pragma unroll
            for (int c = 0; c < 2; ++c) mm1(W, PR * c, cfh[c], cfl[c], net);
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const int wb = 6 * PR * blk;
                hh = ctab_regs(blk);
                dense(wb + 2 * PR, net, hh);                    // hh = fc_0(relu(net)) + b0
                // every term of the residual stream is an accumulation: the next block's fc_c (lines + yz part) goes here, where
                // it covers the conversion of hh
                if (blk + 1 < NBLK) {
                    add_line(blk + 1, gz, sel_z, net);
                    add_line(NBLK + blk + 1, gy, sel_y, net);
#pragma unroll
                    for (int c = 0; c < 2; ++c) mm1(W, wb + 6 * PR + PR * c, cfh[c], cfl[c], net);
                } else {
                    add_line(2 * NBLK, gy, sel_y, net);         // bias of the last fc_1
                }
                dense(wb + 4 * PR, hh, net);                    // net += fc_1(relu(hh))
            }
            f32x16 o = ctab_regs(NBLK);
            dense(30 * PR, net, o);                             // fc_out(relu(net))
            if (hi == 0) store_head(a, hsel, gidx, o[0], o[1], o[2], o[3]);
            if (pool == 0 && ntiles_done < 3) DEC_T(5 + ntiles_done);   // (diagnostic builds: end of this wave's 1st / 2nd / 3rd tile)
            ++ntiles_done;
        }
        item = nxt;
    }
    DEC_T(15);
}

#ifdef GIGA_TRACE
}  // namespace giga
extern "C" int giga_debug_dec_trace(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(giga::g_dec_trace), sizeof(long long) * 16 * 16) == hipSuccess ? 0 : -10;
}
namespace giga {
#endif

// =============================== exact fp32 MFMA path ==============================================
// Same chain on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain).  B operand of MFMA s of a hidden
// layer is simply relu(D[s]) of the previous layer: no conversion, no data movement.
template <int T, bool LATTICE>
__global__ __launch_bounds__(256, 1) void decoder_f32_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // few point batches (e.g. train_giga's single grasp query per scene): heads are spread over blockIdx.y
    const int h_begin = blockIdx.y * a.heads_per_wg;
    const int h_end = h_begin + a.heads_per_wg < a.nheads ? h_begin + a.heads_per_wg : a.nheads;
    const int n = lane & 31, hi = lane >> 5;
    const float4* W = reinterpret_cast<const float4*>(smem);
    const float* ctab = reinterpret_cast<const float*>(smem + (size_t)DEC32_FRAGS * FRAG);
    const float* planes = reinterpret_cast<const float*>(a.planes);
    const size_t plane_stride = LATTICE ? (size_t)a.B * a.R * a.R * CD : (size_t)a.B * RES * RES * CD;

    // The 111 KiB image of the first head is requested before anything else, so the fill (≈5 us at the ≈11 B/clk
    // a CU gets when every workgroup is in its prologue) runs under the feature gather; an image stays resident
    // across point batches while the head does not change (single-head launches: the occupancy path).
    int loaded = -1;
    if ((int)blockIdx.x < a.nbatch && h_begin < h_end) {
        dma_head_image<4, (int)(DEC32_BYTES / FRAG)>(a.blob + a.head_off[h_begin], smem, wave, lane);
        loaded = h_begin;
    }
    bool fresh = true;                                        // the resident image has not been waited for yet

    for (int bseq = blockIdx.x; bseq < a.nbatch; bseq += gridDim.x) {
        const int batch = xcd_swizzle(bseq, a.nbatch);        // an XCD works on a contiguous range of points (L2 locality)
        float cf[T][48], ax0[T], ax1[T];
        long long gidx[T];
        bool valid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long g = ((long long)batch * 4 * T + wave * T + t) * 32 + n;
            valid[t] = g < a.P;
            if (!valid[t]) g = a.P - 1;
            gidx[t] = g;
            int b, r;
            split_scene(g, a.N, a.invN, b, r);
            if constexpr (LATTICE) {
                // lattice-resampled planes: three pixels, no interpolation (see lattice_resample_kernel)
                const int R = a.R, R2 = R * R;
                const int ix = div_magic(r, a.mR2), rz = r - ix * R2;
                const int iy = div_magic(rz, a.mR), iz = rz - iy * R;
                const float px = a.lin[ix], py = a.lin[iy], pz = a.lin[iz];
                ax0[t] = hi ? py : px;
                ax1[t] = hi ? 1.0f : pz;
                const int off[3] = {iz * R + ix, iy * R + ix, iz * R + iy};
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const float* base = planes + pl * plane_stride + ((size_t)b * R2 + off[pl]) * CD + 16 * hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(base + 4 * q);
                        cf[t][16 * pl + 4 * q + 0] = v.x; cf[t][16 * pl + 4 * q + 1] = v.y;
                        cf[t][16 * pl + 4 * q + 2] = v.z; cf[t][16 * pl + 4 * q + 3] = v.w;
                    }
                }
            } else {
            // ---------------- generic gather, line-friendly.  A load instruction in which every lane fetches 16 B of
            // its OWN point touches 64 different cache lines; the gather of a 256-point batch then took a third of the
            // kernel (the CU's texture path handles a few lines per clock).  Here four adjacent lanes fetch the four
            // 16-B quads of one (point, channel half), i.e. one whole 64-B line per four lanes, interpolate them, and
            // the tile is transposed to the operand layout (lane = point) through a wave-private LDS stage.
            const float pxo = a.p[3 * g + 0], pyo = a.p[3 * g + 1], pzo = a.p[3 * g + 2];
            ax0[t] = hi ? pyo : pxo;     // aux MFMA 0: slots (px, py)
            ax1[t] = hi ? 1.0f : pzo;    // aux MFMA 1: slots (pz, 1)
            float* S = reinterpret_cast<float*>(smem + DEC32_BYTES) + wave * (32 * 96);      // [point][96]
            const long long tile0 = ((long long)batch * 4 * T + wave * T + t) * 32;
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const int pair = part * 16 + (lane >> 2), pt = pair >> 1, hf = pair & 1, qd = lane & 3;
                long long gp = tile0 + pt;
                if (gp >= a.P) gp = a.P - 1;
                int bp, rp;
                split_scene(gp, a.N, a.invN, bp, rp);
                const float nx = norm_coord(a.p[3 * gp + 0]), ny = norm_coord(a.p[3 * gp + 1]), nz = norm_coord(a.p[3 * gp + 2]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    // xz: (u,v)=(x,z)  xy: (x,y)  yz: (y,z)      common.py:246-251
                    const Bilin bl = bilin_setup(pl == 2 ? ny : nx, pl == 1 ? ny : nz);
                    const float* base = planes + pl * plane_stride + (size_t)bp * RES * RES * CD + 16 * hf + 4 * qd;
                    const float4 v00 = *reinterpret_cast<const float4*>(base + (size_t)bl.o00 * CD);
                    const float4 v01 = *reinterpret_cast<const float4*>(base + (size_t)bl.o01 * CD);
                    const float4 v10 = *reinterpret_cast<const float4*>(base + (size_t)bl.o10 * CD);
                    const float4 v11 = *reinterpret_cast<const float4*>(base + (size_t)bl.o11 * CD);
                    // same order as aten's grid_sampler: nw, ne, sw, se
                    float4 o;
                    o.x = fmaf(v11.x, bl.w11, fmaf(v10.x, bl.w10, fmaf(v01.x, bl.w01, v00.x * bl.w00)));
                    o.y = fmaf(v11.y, bl.w11, fmaf(v10.y, bl.w10, fmaf(v01.y, bl.w01, v00.y * bl.w00)));
                    o.z = fmaf(v11.z, bl.w11, fmaf(v10.z, bl.w10, fmaf(v01.z, bl.w01, v00.z * bl.w00)));
                    o.w = fmaf(v11.w, bl.w11, fmaf(v10.w, bl.w10, fmaf(v01.w, bl.w01, v00.w * bl.w00)));
                    *reinterpret_cast<float4*>(S + pt * 96 + pl * 32 + 16 * hf + 4 * qd) = o;
                }
            }
            // wave-private stage: DS operations of one wave execute in order, no barrier needed
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(S + n * 96 + pl * 32 + 16 * hi + 4 * q);
                    cf[t][16 * pl + 4 * q + 0] = v.x; cf[t][16 * pl + 4 * q + 1] = v.y;
                    cf[t][16 * pl + 4 * q + 2] = v.z; cf[t][16 * pl + 4 * q + 3] = v.w;
                }
            }
        }
        for (int h = h_begin; h < h_end; ++h) {
            if (loaded != h) {
                __syncthreads();                              // everyone left the previous weight image
                dma_head_image<4, (int)(DEC32_BYTES / FRAG)>(a.blob + a.head_off[h], smem, wave, lane);
                loaded = h;
                fresh = true;
            }
            if (fresh) {
                __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): LDS-DMA landed (builtin: see f16 kernel)
                __syncthreads();
                fresh = false;
            }
            f32x16 net[T], hh[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) net[t][r] = 0.f;
            int k = 0;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        net[t] = mfma32(A.x, cf[t][4 * q + 0], net[t]);
                        net[t] = mfma32(A.y, cf[t][4 * q + 1], net[t]);
                        net[t] = mfma32(A.z, cf[t][4 * q + 2], net[t]);
                        net[t] = mfma32(A.w, cf[t][4 * q + 3], net[t]);
                    }
                }
                {
                    const float4 A = W[(k++) * 64 + lane];
                    const float one0 = hi ? 0.0f : 1.0f;       // aux MFMA 2: slots (1, 0)
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        net[t] = mfma32(A.x, ax0[t], net[t]);
                        net[t] = mfma32(A.y, ax1[t], net[t]);
                        net[t] = mfma32(A.z, one0, net[t]);
                    }
                }
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + blk * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) hh[t] = c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        hh[t] = mfma32(A.x, relu(net[t][4 * q + 0]), hh[t]);
                        hh[t] = mfma32(A.y, relu(net[t][4 * q + 1]), hh[t]);
                        hh[t] = mfma32(A.z, relu(net[t][4 * q + 2]), hh[t]);
                        hh[t] = mfma32(A.w, relu(net[t][4 * q + 3]), hh[t]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        // B operands must be the PRE-update hidden values: hh is not modified here
                        net[t] = mfma32(A.x, relu(hh[t][4 * q + 0]), net[t]);
                        net[t] = mfma32(A.y, relu(hh[t][4 * q + 1]), net[t]);
                        net[t] = mfma32(A.z, relu(hh[t][4 * q + 2]), net[t]);
                        net[t] = mfma32(A.w, relu(hh[t][4 * q + 3]), net[t]);
                    }
                }
            }
            {
                const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) net[t] = mfma32(A.y, ax1[t], net[t]);   // + b1 of block 4 (slot "1.0")
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + NBLK * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
                f32x16 o[T];
#pragma unroll
                for (int t = 0; t < T; ++t) o[t] = c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        o[t] = mfma32(A.x, relu(net[t][4 * q + 0]), o[t]);
                        o[t] = mfma32(A.y, relu(net[t][4 * q + 1]), o[t]);
                        o[t] = mfma32(A.z, relu(net[t][4 * q + 2]), o[t]);
                        o[t] = mfma32(A.w, relu(net[t][4 * q + 3]), o[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (hi == 0 && valid[t]) store_head(a, h, gidx[t], o[t][0], o[t][1], o[t][2], o[t][3]);
            }
        }
    }
}

// ------------------------------- exact fp32 path for a HANDFUL of points --------------------------------
// train_giga's literal call has ONE grasp query per scene (scripts/train_giga.py:141-151: pos is (B, 1, 3)): 32 point-heads per
// head at 32 scenes, i.e. one 32-point tile.  In decoder_f32_kernel that tile is a serial chain of 440 v_mfma_f32_32x32x2_f32 on
// ONE wave (28 k clocks) behind a 111-KiB weight fill: 22 us per launch for 0.1 % of the step's FLOPs.  Here a workgroup of four
// waves takes ONE tile of ONE head and splits the chain where it is not a chain: the contribution of the plane features and of the
// query point to block b's stream, FC_b = fc_c[b](c) + aux_b (12 + 1 fragments, 51 MFMAs), does not depend on the stream, so the
// four waves compute FC_0 .. FC_4 side by side (wave 1 takes two of them) while wave 0 walks the 5 x 32 + 5 dependent MFMAs of
// fc_0 / fc_1 / fc_out and adds FC_b when it reaches block b.  The FC fragments are read straight from the packed blob (global /
// L2, one 16-byte load per lane and fragment, all of a block in flight at once); the 45 fragments of the dependent chain are
// copied into LDS by waves 1-3 (LDS-DMA) while wave 0 computes FC_0 (read from global inside the chain they cost a load latency
// each: the compiler sinks every load to its use).  Same fragments, same k-slot maps and
// the same operands as decoder_f32_kernel; the stream is summed as (stream + FC_b) instead of one running fma chain, an fp32
// rounding-level difference (1e-7 relative).
__global__ __launch_bounds__(256) void decoder_f32_tile_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y;
    const long long tile0 = (long long)blockIdx.x * 32;
    const float4* W = reinterpret_cast<const float4*>(a.blob + a.head_off[h]);          // fragments in global memory
    const float* ctab = reinterpret_cast<const float*>(a.blob + a.head_off[h] + (size_t)DEC32_FRAGS * FRAG);
    const float* planes = reinterpret_cast<const float*>(a.planes);
    const size_t plane_stride = (size_t)a.B * RES * RES * CD;
    constexpr int BLKF = 21;                                   // fragments per block: 12 features, 1 aux, 4 fc_0, 4 fc_1
    // which FC blocks this wave computes: wave 0 -> 0, wave 1 -> 1 and 4, wave 2 -> 2, wave 3 -> 3
    const int blk_a = wave, blk_b = wave == 1 ? 4 : -1;
    // ---- waves 1-3: the chain's fragments (fc_0, fc_1 of the five blocks, the tail, the C table) -> LDS slots 0 .. 45
    uint8_t* CHW = smem + 4 * 32 * 96 * sizeof(float) + NBLK * 4 * 64 * sizeof(float4);
    if (wave != 0) {
        const uint8_t* hb = a.blob + a.head_off[h] + lane * 16;
        for (int c = wave - 1; c < 46; c += 3) {
            const int frag = c < 40 ? BLKF * (c >> 3) + 13 + (c & 7) : BLKF * NBLK + (c - 40);     // (c = 45: the C table chunk)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hb + (size_t)frag * FRAG),
                                             (__attribute__((address_space(3))) void*)(CHW + c * FRAG), 16, 0, 0);
        }
    }
    // ---- the first FC block's fragments are requested before the gather (they do not depend on it)
    float4 fa[13];
#pragma unroll
    for (int q = 0; q < 13; ++q) fa[q] = W[(size_t)(BLKF * blk_a + q) * 64 + lane];
    // ---- gather: every wave gathers the tile for itself (decoder_f32_kernel's line-friendly gather, wave-private LDS stage)
    long long g = tile0 + n;
    const bool valid = g < a.P;
    if (!valid) g = a.P - 1;
    float cf[48];
    const float ax0 = hi ? a.p[3 * g + 1] : a.p[3 * g + 0];     // aux MFMA 0: slots (px, py)
    const float ax1 = hi ? 1.0f : a.p[3 * g + 2];               // aux MFMA 1: slots (pz, 1)
    {
        float* S = reinterpret_cast<float*>(smem) + wave * (32 * 96);      // [point][96]
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            const int pair = part * 16 + (lane >> 2), pt = pair >> 1, hf = pair & 1, qd = lane & 3;
            long long gp = tile0 + pt;
            if (gp >= a.P) gp = a.P - 1;
            int bp, rp;
            split_scene(gp, a.N, a.invN, bp, rp);
            const float nx = norm_coord(a.p[3 * gp + 0]), ny = norm_coord(a.p[3 * gp + 1]), nz = norm_coord(a.p[3 * gp + 2]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const Bilin bl = bilin_setup(pl == 2 ? ny : nx, pl == 1 ? ny : nz);
                const float* base = planes + pl * plane_stride + (size_t)bp * RES * RES * CD + 16 * hf + 4 * qd;
                const float4 v00 = *reinterpret_cast<const float4*>(base + (size_t)bl.o00 * CD);
                const float4 v01 = *reinterpret_cast<const float4*>(base + (size_t)bl.o01 * CD);
                const float4 v10 = *reinterpret_cast<const float4*>(base + (size_t)bl.o10 * CD);
                const float4 v11 = *reinterpret_cast<const float4*>(base + (size_t)bl.o11 * CD);
                float4 o;
                o.x = fmaf(v11.x, bl.w11, fmaf(v10.x, bl.w10, fmaf(v01.x, bl.w01, v00.x * bl.w00)));
                o.y = fmaf(v11.y, bl.w11, fmaf(v10.y, bl.w10, fmaf(v01.y, bl.w01, v00.y * bl.w00)));
                o.z = fmaf(v11.z, bl.w11, fmaf(v10.z, bl.w10, fmaf(v01.z, bl.w01, v00.z * bl.w00)));
                o.w = fmaf(v11.w, bl.w11, fmaf(v10.w, bl.w10, fmaf(v01.w, bl.w01, v00.w * bl.w00)));
                *reinterpret_cast<float4*>(S + pt * 96 + pl * 32 + 16 * hf + 4 * qd) = o;
            }
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(S + n * 96 + pl * 32 + 16 * hi + 4 * q);
                cf[16 * pl + 4 * q + 0] = v.x; cf[16 * pl + 4 * q + 1] = v.y; cf[16 * pl + 4 * q + 2] = v.z; cf[16 * pl + 4 * q + 3] = v.w;
            }
    }
    const float one0 = hi ? 0.0f : 1.0f;                        // aux MFMA 2: slots (1, 0)
    auto fc_block = [&](const float4 (&A)[13]) {                // FC_b: 12 feature fragments + the aux fragment, from zero
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            acc = mfma32(A[q].x, cf[4 * q + 0], acc);
            acc = mfma32(A[q].y, cf[4 * q + 1], acc);
            acc = mfma32(A[q].z, cf[4 * q + 2], acc);
            acc = mfma32(A[q].w, cf[4 * q + 3], acc);
        }
        acc = mfma32(A[12].x, ax0, acc);
        acc = mfma32(A[12].y, ax1, acc);
        acc = mfma32(A[12].z, one0, acc);
        return acc;
    };
    // FC exchange area behind the four gather stages: FC_b as [4 quads][64 lanes] float4
    float4* FCX = reinterpret_cast<float4*>(smem + 4 * 32 * 96 * sizeof(float));
    auto put_fc = [&](int blk, const f32x16& v) {
#pragma unroll
        for (int q = 0; q < 4; ++q) FCX[(blk * 4 + q) * 64 + lane] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    };
    f32x16 net = fc_block(fa);                                  // FC of this wave's first block
    if (wave != 0) put_fc(blk_a, net);
    if (blk_b >= 0) {                                           // (wave 1) the second FC block's fragments
#pragma unroll
        for (int q = 0; q < 13; ++q) fa[q] = W[(size_t)(BLKF * blk_b + q) * 64 + lane];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): this wave's share of the chain fragments has landed
    __syncthreads();                                            // FC_1 .. FC_3 and the chain's fragments are in LDS
    if (blk_b >= 0) put_fc(blk_b, fc_block(fa));
    const float4* CW = reinterpret_cast<const float4*>(CHW);
    const float* ctab_l = reinterpret_cast<const float*>(CHW + 45 * FRAG);
    auto chain_block = [&](int blk) __attribute__((always_inline)) {
            if (blk > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = FCX[(blk * 4 + q) * 64 + lane];
                    net[4 * q] += v.x; net[4 * q + 1] += v.y; net[4 * q + 2] += v.z; net[4 * q + 3] += v.w;
                }
            }
            f32x16 hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ctab_l + blk * 32 + 8 * q + 4 * hi);
                hh[4 * q + 0] = v.x; hh[4 * q + 1] = v.y; hh[4 * q + 2] = v.z; hh[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 A = CW[(blk * 8 + q) * 64 + lane];
                hh = mfma32(A.x, relu(net[4 * q + 0]), hh);
                hh = mfma32(A.y, relu(net[4 * q + 1]), hh);
                hh = mfma32(A.z, relu(net[4 * q + 2]), hh);
                hh = mfma32(A.w, relu(net[4 * q + 3]), hh);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 A = CW[(blk * 8 + 4 + q) * 64 + lane];
                net = mfma32(A.x, relu(hh[4 * q + 0]), net);
                net = mfma32(A.y, relu(hh[4 * q + 1]), net);
                net = mfma32(A.z, relu(hh[4 * q + 2]), net);
                net = mfma32(A.w, relu(hh[4 * q + 3]), net);
            }
    };
    // Wave 0 runs the chain; its block 4 needs FC_4, which wave 1 computes after the first barrier.  The second barrier sits in
    // UNIFORM control flow (every wave reaches the same __syncthreads): blocks 0..3, barrier, block 4 and the tail.
    if (wave == 0) {
#pragma unroll
        for (int blk = 0; blk < NBLK - 1; ++blk) chain_block(blk);
    }
    __syncthreads();                                            // FC_4 (wave 1's second block) is in LDS
    if (wave == 0) {
        chain_block(NBLK - 1);
        net = mfma32(CW[40 * 64 + lane].y, ax1, net);           // + b1 of block 4 (slot "1.0")
        f32x16 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(ctab_l + NBLK * 32 + 8 * q + 4 * hi);
            o[4 * q + 0] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 A = CW[(41 + q) * 64 + lane];
            o = mfma32(A.x, relu(net[4 * q + 0]), o);
            o = mfma32(A.y, relu(net[4 * q + 1]), o);
            o = mfma32(A.z, relu(net[4 * q + 2]), o);
            o = mfma32(A.w, relu(net[4 * q + 3]), o);
        }
        if (hi == 0 && valid) store_head(a, h, g, o[0], o[1], o[2], o[3]);
    }
}
constexpr size_t DEC32_TILE_LDS = 4 * 32 * 96 * sizeof(float) + NBLK * 4 * 64 * sizeof(float4) + 46 * FRAG;   // 4 gather stages, FC exchange, chain fragments

// ------------------------------- plane repack NCHW fp32 -> NHWC T ---------------------------------
// Used when the planes arrive through the Python boundary as the reference's (B,32,40,40) tensors
// (LocalDecoder.forward(p, c_plane), decoder.py:133).  One thread per (image, pixel, 4 channels).
template <typename TOut>
__global__ void planes_nchw_to_nhwc_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                           const float* __restrict__ src2, TOut* __restrict__ dst, int B) {
    __shared__ float tile[32][RES + 1];
    // block = one (plane, scene, row y): transposes a 32(c) x 40(x) slab
    const int img = blockIdx.x / RES, y = blockIdx.x % RES;
    const int pl = img / B, b = img % B;
    const float* src = (pl == 0 ? src0 : pl == 1 ? src1 : src2) + (size_t)b * CD * RES * RES;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int c = i / RES, x = i % RES;
        tile[c][x] = src[(size_t)c * RES * RES + y * RES + x];
    }
    __syncthreads();
    TOut* d = dst + ((size_t)img * RES * RES + (size_t)y * RES) * CD;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int x = i / CD, c = i % CD;
        d[(size_t)x * CD + c] = (TOut)tile[c][x];
    }
}

template <typename TIn>
__global__ void planes_nhwc_to_nchw_kernel(const TIn* __restrict__ src, float* __restrict__ dst) {
    __shared__ float tile[32][RES + 1];
    const int img = blockIdx.x / RES, y = blockIdx.x % RES;
    const TIn* s = src + ((size_t)img * RES * RES + (size_t)y * RES) * CD;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int x = i / CD, c = i % CD;
        tile[c][x] = (float)s[(size_t)x * CD + c];
    }
    __syncthreads();
    float* d = dst + (size_t)img * CD * RES * RES;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int c = i / RES, x = i % RES;
        d[(size_t)c * RES * RES + y * RES + x] = tile[c][x];
    }
}

// ------------------------------- lattice resampling --------------------------------------------------
// Inference queries the fixed R^3 lattice (detection_implicit.py:28-31), so every plane is sampled at
// only R*R distinct positions, each shared by R points.  One thread per (plane, scene, j, i, 8
// channels) evaluates sample_plane_feature (decoder.py:117-122) once; out [3][B][R(v)][R(u)][32].
// (TIn = float, TP = half: fp32 planes of an fp32 / f16x3 encoder feeding the plain-f16 decoder -- GIGA_PLANES_FP32)
template <typename TP, typename TIn = TP>
__global__ void lattice_resample_kernel(const TIn* __restrict__ planes, const float* __restrict__ lin,
                                        TP* __restrict__ out, int B, int R) {
    const long long total = 3LL * B * R * R * 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i & 3);
    const long long pix = i >> 2;
    const int iu = (int)(pix % R), iv = (int)((pix / R) % R);
    const long long img = pix / ((long long)R * R);
    const Bilin bl = bilin_setup(norm_coord(lin[iu]), norm_coord(lin[iv]));
    const TIn* src = planes + (size_t)img * RES * RES * CD + 8 * cg;
    TP* dst = out + (size_t)pix * CD + 8 * cg;
    // the thread's 8 channels of a tap are 16 (f16) or 32 (fp32) contiguous bytes: one or two 16-byte loads per tap, one or two
    // 16-byte stores (element-wise 2-byte accesses made this kernel 14 us at 32 scenes for 17 MB)
    typedef TP vec8 __attribute__((ext_vector_type(8)));
    typedef TIn vin8 __attribute__((ext_vector_type(8)));
    const vin8 v00 = *reinterpret_cast<const vin8*>(src + (size_t)bl.o00 * CD);
    const vin8 v01 = *reinterpret_cast<const vin8*>(src + (size_t)bl.o01 * CD);
    const vin8 v10 = *reinterpret_cast<const vin8*>(src + (size_t)bl.o10 * CD);
    const vin8 v11 = *reinterpret_cast<const vin8*>(src + (size_t)bl.o11 * CD);
    vec8 r;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float acc = (float)v00[c] * bl.w00;
        acc = fmaf((float)v01[c], bl.w01, acc);
        acc = fmaf((float)v10[c], bl.w10, acc);
        acc = fmaf((float)v11[c], bl.w11, acc);
        r[c] = (TP)acc;
    }
    *reinterpret_cast<vec8*>(dst) = r;
}


// f16x3 variant: fp32 planes in, interpolated in fp32, written as split pixels: 4 groups of 8 channels x
// [8 hi halfs | 8 lo halfs] (128 B per pixel, the same bytes as fp32) -- the decoder's B operands, ready-made.
__global__ void lattice_resample_split_kernel(const float* __restrict__ planes, const float* __restrict__ lin,
                                              half_t* __restrict__ out, int B, int R) {
    const long long total = 3LL * B * R * R * 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i & 3);
    const long long pix = i >> 2;
    const int iu = (int)(pix % R), iv = (int)((pix / R) % R);
    const long long img = pix / ((long long)R * R);
    const Bilin bl = bilin_setup(norm_coord(lin[iu]), norm_coord(lin[iv]));
    const float* src = planes + (size_t)img * RES * RES * CD + 8 * cg;
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 v00 = *reinterpret_cast<const float4*>(src + (size_t)bl.o00 * CD + 4 * q);
        const float4 v01 = *reinterpret_cast<const float4*>(src + (size_t)bl.o01 * CD + 4 * q);
        const float4 v10 = *reinterpret_cast<const float4*>(src + (size_t)bl.o10 * CD + 4 * q);
        const float4 v11 = *reinterpret_cast<const float4*>(src + (size_t)bl.o11 * CD + 4 * q);
        v[4 * q + 0] = fmaf(v11.x, bl.w11, fmaf(v10.x, bl.w10, fmaf(v01.x, bl.w01, v00.x * bl.w00)));
        v[4 * q + 1] = fmaf(v11.y, bl.w11, fmaf(v10.y, bl.w10, fmaf(v01.y, bl.w01, v00.y * bl.w00)));
        v[4 * q + 2] = fmaf(v11.z, bl.w11, fmaf(v10.z, bl.w10, fmaf(v01.z, bl.w01, v00.z * bl.w00)));
        v[4 * q + 3] = fmaf(v11.w, bl.w11, fmaf(v10.w, bl.w10, fmaf(v01.w, bl.w01, v00.w * bl.w00)));
    }
    half8 h, l;
    split8(v, h, l);
    half_t* dst = out + (size_t)pix * 2 * CD + 16 * cg;
    *reinterpret_cast<half8*>(dst) = h;
    *reinterpret_cast<half8*>(dst + 8) = l;
}

// ------------------------------- launchers ----------------------------------------------------------
constexpr size_t DEC32_LDS = DEC32_BYTES + 4 * 32 * 96 * sizeof(float);     // weight image + the gather's per-wave stage
static_assert(DEC32_LDS <= 160 * 1024, "LDS budget of the fp32 decoder");
int launch_decoder(const DecArgs& a0, int precision, hipStream_t s, void* ev0, void* ev1) {
    DecArgs a = a0;
    if (a.P <= 0 || a.nheads <= 0) return 0;
    if (ev0 && ev1) (void)hipEventRecord(static_cast<hipEvent_t>(ev0), s);
    const long long tiles = (a.P + 31) / 32;
    const bool lat = a.R > 0;
    a.invN = 1.0f / (float)a.N;
    if (lat) {
        a.mR = (unsigned)((0x100000000ULL + a.R - 1) / a.R);
        a.mR2 = (unsigned)((0x100000000ULL + (unsigned long long)a.R * a.R - 1) / ((unsigned long long)a.R * a.R));
    }
    // Head-resident kernel: always for the f16x3 split mode; for plain f16 when the launch is small (the shared-feature
    // kernel below walks rounds of NW*T tiles per workgroup with all heads on the same CU: a single scene's 2000 tiles then
    // keep only 84 CUs busy: 31.6 -> 16.3 us for the three grasp heads of one scene's 64 000 lattice points; from ~8 scenes
    // up the shared-feature kernel is the faster one, 297 vs 323 us at 32 scenes).  grid = slots x nheads <= 256 (one workgroup per CU).  Two tiles per wave (the weight fragments
    // are read from LDS once per two tiles) when every workgroup gets at least two such rounds, else one tile per wave.
    // (tuning knob, read once: GIGA_DEC16_RESIDENT=0/1 forces the choice for plain f16; measurements in DESIGN.md)
    // Lattice launches of the f16-class modes with R % 8 == 0 (the inference lattice: R = 40): the separable-fc_c kernel
    // (decoder_lat_kernel) -- f16x3 always (1 scene 31.8 -> 27.8 us, 32 scenes 0.754 -> 0.583 ms), plain f16 from 4 scenes up (32 scenes
    // 0.279 -> 0.260 ms; a single scene is latency-bound and 1 us faster on the head-resident kernel below, which also serves other
    // lattices).  (tuning knob: GIGA_DEC_LAT=0 / 1 forces the choice)
    const int force_lat = [] { const char* e = getenv("GIGA_DEC_LAT"); return e ? atoi(e) : -1; }();   // (per call: A/B runs in one process)
    if (lat && (precision == 1 || precision == 2) && a.R <= LAT_MAX_R && a.R % 8 == 0 && (force_lat >= 0 ? force_lat != 0 : (precision == 2 || a.B >= 4))) {
        const int cap = 256 / a.nheads, cap8 = 256 / (8 * a.nheads) * 8;
        // small launches: hand a slab out in parts (NP divides the tile count), so that a single scene still covers the chip; a
        // part costs its tiles plus one line phase (~ the time of 12 tiles on a workgroup)
        const int ntile = (a.R / 4) * (a.R / 8);
        int np = 1;
        {
            double best = 1e30;
            for (int c : {1, 2, 5}) {
                if (ntile % c) continue;
                const long long units = (long long)a.B * a.R * c;
                const long long rounds = (units + cap - 1) / cap;
                const double cost = (double)rounds * ((double)ntile / c + 12.0);
                if (cost < best - 1e-9) { best = cost; np = c; }
            }
        }
        a.lat_parts = np;
        const int nslab = a.B * a.R * np;
        int slots = nslab < cap ? nslab : (nslab >= 2 * cap8 && cap8 > 0 ? cap8 : cap);
        a.nbatch = slots;
        // (tuning knob: GIGA_LAT_NW = 12 | 16 waves per workgroup; 16 = four per SIMD, the default: 0.267 -> 0.257 ms at 32 scenes)
        const int nw = [] { const char* e = getenv("GIGA_LAT_NW"); return e ? atoi(e) : 16; }();
        auto go = [&](auto kern, int NWv, size_t lds) {
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
            GIGA_LAUNCH(kern, dim3(slots * a.nheads), dim3(NWv * 64), lds, s, a);
        };
        if (precision == 2) {
            if (nw == 16) go(decoder_lat_kernel<true, 16, false>, 16, lat_lds_bytes<true, false>(a.R));
            else go(decoder_lat_kernel<true, 12, false>, 12, lat_lds_bytes<true, false>(a.R));
        } else {
            if (nw == 16) go(decoder_lat_kernel<false, 16, true>, 16, lat_lds_bytes<false, true>(a.R));
            else go(decoder_lat_kernel<false, 12, true>, 12, lat_lds_bytes<false, true>(a.R));
        }
        if (ev0 && ev1) (void)hipEventRecord(static_cast<hipEvent_t>(ev1), s);
        return hipGetLastError() == hipSuccess ? 0 : -10;
    }
    static const int force_resident = [] { const char* e = getenv("GIGA_DEC16_RESIDENT"); return e ? atoi(e) : -1; }();
    const bool resident = precision == 2 ||
                          (precision == 1 && (force_resident >= 0 ? force_resident != 0 : tiles * a.nheads < (lat ? 16000 : 6000)));
    if (resident) {
        const int cap = 256 / a.nheads;                        // one workgroup per CU
        const int cap8 = 256 / (8 * a.nheads) * 8;             // ... or a multiple of 8 slots: XCD-contiguous ranges (L2 locality)
        auto go = [&](auto kern, int NW, int T, size_t lds) {
            const int per_round = NW * T;
            int slots = (int)((tiles + per_round - 1) / per_round);
            // large launches trade the last few CUs for L2 locality; small ones (a single scene: 2000 tiles) are latency-bound and
            // take every CU -- 240 instead of 255 workgroups would add a third, nearly empty round
            if (slots > cap) slots = tiles >= 8LL * cap * per_round ? cap8 : cap;
            else if (slots >= 8 && tiles >= 8LL * slots) slots &= ~7;
            a.nbatch = slots;
            giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)lds);
            GIGA_LAUNCH(kern, dim3(slots * a.nheads), dim3(NW * 64), lds, s, a);
        };
        if (precision == 2) {
            constexpr int NW = 8;
            const bool two = tiles >= 2LL * cap * NW * 2;
            if (two) { if (lat) go(decoder_f16s_kernel<2, true, NW, true>, NW, 2, DEC16S_BYTES); else go(decoder_f16s_kernel<2, false, NW, true>, NW, 2, DEC16S_BYTES); }
            else { if (lat) go(decoder_f16s_kernel<1, true, NW, true>, NW, 1, DEC16S_BYTES); else go(decoder_f16s_kernel<1, false, NW, true>, NW, 1, DEC16S_BYTES); }
        } else {
            // plain f16: 112-142 VGPRs with one tile per wave (12 waves), 2 tiles need the 256-register budget (8 waves)
            const bool two = tiles >= 2LL * cap * 8 * 2;
            if (two) { if (lat) go(decoder_f16s_kernel<2, true, 8, false>, 8, 2, DEC16_BYTES); else go(decoder_f16s_kernel<2, false, 8, false>, 8, 2, DEC16_BYTES); }
            else { if (lat) go(decoder_f16s_kernel<1, true, 12, false>, 12, 1, DEC16_BYTES); else go(decoder_f16s_kernel<1, false, 12, false>, 12, 1, DEC16_BYTES); }
        }
    } else if (precision == 1 && lat) {
        // lattice variant: 155 VGPRs -> 12 waves (3 per SIMD, phases 0/1/2 over the heads)
        constexpr int T = 2, NW = 12;
        a.nbatch = (int)((tiles + NW * T - 1) / (NW * T));
        const int grid = a.nbatch < 256 ? a.nbatch : 256;      // one persistent workgroup per CU
        auto kern = decoder_f16_kernel<T, true, NW>;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)(2 * DEC16_BYTES));
        GIGA_LAUNCH(kern, dim3(grid), dim3(NW * 64), 2 * DEC16_BYTES, s, a);
    } else if (precision == 1) {
        constexpr int T = 2, NW = 8;
        a.nbatch = (int)((tiles + NW * T - 1) / (NW * T));
        const int grid = a.nbatch < 256 ? a.nbatch : 256;      // one persistent workgroup per CU
        auto kern = decoder_f16_kernel<T, false, NW>;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)(2 * DEC16_BYTES));
        GIGA_LAUNCH(kern, dim3(grid), dim3(NW * 64), 2 * DEC16_BYTES, s, a);
    } else if (!lat && a.N == 1 && tiles * a.nheads <= 96 && [] { const char* e = getenv("GIGA_DEC32_TILE"); return e ? atoi(e) != 0 : true; }()) {
        // ONE query per scene (train_giga's literal call) and few scenes: one tile and head per workgroup, four waves per chain.
        // (Only for N == 1: the kernel sums the stream in a different order than decoder_f32_kernel, and callers that split a
        // query set into chunks -- Generator3D.eval_points -- rely on a point's result not depending on the chunk it is in.)
        auto kern = decoder_f32_tile_kernel;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)DEC32_TILE_LDS);
        GIGA_LAUNCH(kern, dim3((unsigned)tiles, a.nheads), dim3(256), DEC32_TILE_LDS, s, a);
    } else if (tiles >= 2 * 4 * 256) {
        // enough work for two tiles per wave on every CU: the 111 KiB weight image is staged half as often
        constexpr int T = 2;
        a.nbatch = (int)((tiles + 4 * T - 1) / (4 * T));
        const int grid = a.nbatch < 256 ? a.nbatch : 256;
        auto kern = lat ? decoder_f32_kernel<T, true> : decoder_f32_kernel<T, false>;
        a.heads_per_wg = a.nheads;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)DEC32_LDS);
        GIGA_LAUNCH(kern, dim3(grid), dim3(256), DEC32_LDS, s, a);
    } else {
        constexpr int T = 1;
        a.nbatch = (int)((tiles + 4 * T - 1) / (4 * T));
        const int grid = a.nbatch < 256 ? a.nbatch : 256;
        auto kern = lat ? decoder_f32_kernel<T, true> : decoder_f32_kernel<T, false>;
        // when the point batches cannot fill the chip, give every head its own workgroups
        const bool split = grid * a.nheads <= 256;
        a.heads_per_wg = split ? 1 : a.nheads;
        giga::dyn_lds_once(reinterpret_cast<const void*>(kern), (int)DEC32_LDS);
        GIGA_LAUNCH(kern, dim3(grid, split ? a.nheads : 1), dim3(256), DEC32_LDS, s, a);
    }
    if (ev0 && ev1) (void)hipEventRecord(static_cast<hipEvent_t>(ev1), s);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_lattice_resample(const void* planes, const float* lin, void* out, int B, int R, int precision,
                            hipStream_t s, bool planes_fp32) {
    const long long total = 3LL * B * R * R * 4;
    if (total <= 0) return 0;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (precision == 2)
        GIGA_LAUNCH(lattice_resample_split_kernel, dim3(grid), dim3(256), 0, s,
                           reinterpret_cast<const float*>(planes), lin, reinterpret_cast<half_t*>(out), B, R);
    else if (precision == 1 && planes_fp32)
        GIGA_LAUNCH((lattice_resample_kernel<half_t, float>), dim3(grid), dim3(256), 0, s,
                           reinterpret_cast<const float*>(planes), lin, reinterpret_cast<half_t*>(out), B, R);
    else if (precision == 1)
        GIGA_LAUNCH(lattice_resample_kernel<half_t>, dim3(grid), dim3(256), 0, s,
                           reinterpret_cast<const half_t*>(planes), lin, reinterpret_cast<half_t*>(out), B, R);
    else
        GIGA_LAUNCH(lattice_resample_kernel<float>, dim3(grid), dim3(256), 0, s,
                           reinterpret_cast<const float*>(planes), lin, reinterpret_cast<float*>(out), B, R);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_planes_pack(const float* xz, const float* xy, const float* yz, void* dst, int B, int precision,
                       hipStream_t s) {
    if (B <= 0) return 0;
    if (precision == 1)
        GIGA_LAUNCH(planes_nchw_to_nhwc_kernel<half_t>, dim3(3 * B * RES), dim3(256), 0, s, xz, xy, yz,
                           reinterpret_cast<half_t*>(dst), B);
    else
        GIGA_LAUNCH(planes_nchw_to_nhwc_kernel<float>, dim3(3 * B * RES), dim3(256), 0, s, xz, xy, yz,
                           reinterpret_cast<float*>(dst), B);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_planes_unpack(const void* src, float* dst, int B, int precision, hipStream_t s) {
    if (B <= 0) return 0;
    if (precision == 1)
        GIGA_LAUNCH(planes_nhwc_to_nchw_kernel<half_t>, dim3(3 * B * RES), dim3(256), 0, s,
                           reinterpret_cast<const half_t*>(src), dst);
    else
        GIGA_LAUNCH(planes_nhwc_to_nchw_kernel<float>, dim3(3 * B * RES), dim3(256), 0, s,
                           reinterpret_cast<const float*>(src), dst);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
