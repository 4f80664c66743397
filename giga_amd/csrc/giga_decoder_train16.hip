// bf16 decoder of the bf16 TRAINING step (BASELINE config c5; scripts/train_giga.py:198-211): LocalDecoder.forward
// (conv_onet/models/decoder.py:133-176, ResnetBlockFC layers.py:39-47) and the gradient of every one of its parameters and of
// the feature planes, with bf16 MFMA operands and fp32 accumulation -- the arithmetic torch.autocast(bfloat16) gives nn.Linear.
//
// ONE kernel computes a head's backward: no [P][32] row arrays in HBM, no separate weight-gradient launch.
//   dect_kernel<false>  forward: gather (12 bilinear taps per point on the fp32 NHWC planes, rounded to bf16 once), the 11-layer
//                       chain on v_mfma_f32_32x32x16_bf16 in the transposed, lane = point layout of the f16 inference kernel
//                       (weights = A operand out of an LDS image, activations = B operand, fp32 residual stream in the accumulator).
//   dect_kernel<true>   backward: the same forward recomputed (its bf16 activations stay in REGISTERS: 80 VGPRs), then
//                         G = DN[b+1];  DH[b] = (W1_b^T bf16(G)) * (h_b > 0);  DN[b] = G + (W0_b^T bf16(DH[b])) * (net_b > 0);
//                         dc += Wc_b^T bf16(DN[b])
//                       on the MFMA with transposed-matrix fragments, again without cross-lane traffic; and IN THE SAME KERNEL the
//                       weight gradients dW = dY^T X (contraction over the POINTS): every wave parks its bf16 (dY, X) tiles of a
//                       block in LDS ([32 points][32 features], 8-byte pieces swizzled), and after one workgroup barrier the waves
//                       read them back TRANSPOSED with ds_read_b64_tr_b16 -- lane = feature, k-slots = points: exactly the A / B
//                       operands of v_mfma_f32_32x32x16_bf16 -- each wave for the 32 x 32 gradient tiles it owns.  The 31 tiles of
//                       a head (15 fc_c, 5 fc_0, 5 fc_1, fc_out, 5 "bias" tiles whose B operand is the aux tile [p_hi, 1, p_lo] /
//                       a constant one-hot column: column sums = bias gradients, fc_p) stay in accumulator registers across the
//                       workgroup's persistent loop over its points (8 tiles = 128 VGPRs per wave) and leave ONCE per workgroup as
//                       a partial image; dect_reduce_kernel sums the partial images in a fixed order (deterministic) into the flat
//                       gradient buffer.
// Plane gradients: dc leaves as rows [P][96] for plane_gather_kernel (many queries per scene) or is scattered with fp32 atomics
// through an LDS transposition (few queries), as in the fp32 path (giga_decoder_bwd.hip).
// tools/tr16_probe.hip pins the lane map of ds_read_b64_tr_b16 and the tile layout on the device (profiles/r05/tr16_probe.txt).
#include "giga_args.h"
#include "giga_dev.h"
#include "giga_dect.h"

namespace giga {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s4v* lds_s4v;

__device__ __forceinline__ f32x16 mfma_bf(bf8 a, bf8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- gradient tiles of a head ------------------------------------------------------------------------------------------
// tile id 6 b + t: t = 0..2 fc_c[b] columns 32 t .. 32 t + 31, 3 bias tile of block b, 4 fc_1[b], 5 fc_0[b]; 30 fc_out.
// bias tile columns: 0..2 dY x p_hi, 3 colsum(DN[b]), 4..6 dY x p_lo, 8 colsum(DH[b]), 9 colsum(DN[5]) (b = 4), 10 colsum(dO) (b = 4)
constexpr int DECT_NTILES = 6 * NBLK + 1;                       // 31
constexpr int DECT_TILE_FLOATS = 1024;
// wave w plays role (w + b) & 3 in block b: role 0 -> tiles t = 0, 1; role 1 -> t = 2, 3; role 2 -> t = 4; role 3 -> t = 5.
// accumulator slot of the first tile wave w owns in block b (blocks in ascending order); wave 2 also owns fc_out (slot 7)
__host__ __device__ constexpr int dect_role_ntiles(int role) { return role < 2 ? 2 : 1; }
__host__ __device__ constexpr int dect_slot(int w, int b) {
    int s = 0;
    for (int k = 0; k < b; ++k) s += dect_role_ntiles((w + k) & 3);
    return s;
}
__host__ __device__ constexpr int dect_role_tile(int role) { return role == 0 ? 0 : role == 1 ? 2 : role == 2 ? 4 : 5; }
constexpr int DECT_ACC = 8;
static_assert(dect_slot(0, NBLK) == 8 && dect_slot(1, NBLK) == 8 && dect_slot(2, NBLK) == 7 && dect_slot(3, NBLK) == 7, "tile ownership");

// ---- LDS layout ---------------------------------------------------------------------------------------------------------------
// [zone: per wave 2 x (XH, DH, XN) step buffers + 3 DN slots = 9 tiles; the forward image overlays it while the forward chain runs]
// [per wave: C tiles (3), aux tile [8 columns][32 points], dO tile [4][32]] [backward image]
constexpr int TILE_B = 2048;
constexpr int ZONE_WAVE = 9 * TILE_B;                           // 18432
constexpr int ZONE_B = 4 * ZONE_WAVE;                           // 73728 >= DECT_FWD_BYTES
constexpr int CW_B = 3 * TILE_B + 512 + 256;                    // 6912
constexpr int LDS_FWD = ZONE_B + 4 * CW_B;                      // 101376: end of the backward's per-wave regions
constexpr int LDS_BWD = LDS_FWD + (int)DECT_BWD_BYTES;          // 153600
constexpr int FWD_NW = 8;
constexpr int LDS_FWD8 = ZONE_B + FWD_NW * CW_B;                // 129024: forward kernel (image in the zone + 8 waves' C tiles)
static_assert(ZONE_B >= (int)DECT_FWD_BYTES && LDS_BWD <= 160 * 1024, "LDS budget of the bf16 training decoder");

// byte address of the 8-byte piece (point p, features 4q .. 4q+3) of a [32][32] bf16 tile: rows of 64 bytes, the pieces of a row
// permuted by the row PAIR -- the ds_write_b64 of 16 consecutive points with one q and the transposing read of 4 consecutive points
// x 8 pieces both touch every bank once (profiles/r05/tr16_probe.txt)
__device__ __forceinline__ int piece_addr(int p, int q) { return 64 * p + 8 * (q ^ ((p >> 1) & 7)); }

struct DectArgs {
    const float* planes;            // fp32 NHWC [3][B][40][40][32]
    const float* p;                 // [P][3]
    const uint8_t* img_fwd[NHEADS]; // bf16 forward image of requested head k
    const uint8_t* img_bwd[NHEADS]; // bf16 backward image
    int head_id[NHEADS];
    float* out[NHEADS];             // forward: written.  backward: the forward's outputs (post sigmoid / normalize), read
    const float* dout[NHEADS];
    float* partial[NHEADS];         // backward: [gridDim.x][DECT_NTILES][1024] partial gradient tiles
    float* gplanes;                 // plane gradients (atomics) or nullptr (detached head)
    float* dcbuf;                   // [P][96] rows for plane_gather_kernel, or nullptr
    int nheads, B, N, post;
    long long P;
    float invN;
};

// 64 lanes x 16 B chunks of an image by LDS-DMA, the chunks dealt over the workgroup's waves
template <int CHUNKS, int NW>
__device__ __forceinline__ void dect_dma(const uint8_t* src, uint8_t* lds_dst, int wave, int lane) {
    for (int c = wave; c < CHUNKS; c += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * FRAG + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds_dst + c * FRAG), 16, 0, 0);
}

__device__ __forceinline__ bf8 relu_bf8(const f32x16& d, int c) {      // bf16(relu(D registers 8c .. 8c+7)), round to nearest even
    bf8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (__bf16)relu(d[8 * c + j]);
    return x;
}
__device__ __forceinline__ bf8 cvt_bf8(const f32x16& d, int c) {
    bf8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (__bf16)d[8 * c + j];
    return x;
}
// d[8c + j] = x[j] != 0 ? d[8c + j] : 0   (x = bf16(relu(pre-activation)): non-zero exactly where the pre-activation was positive)
__device__ __forceinline__ void mask_by(f32x16& d, int c, bf8 x) {
    const uint4 w = __builtin_bit_cast(uint4, x);
    const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d[8 * c + 2 * k] = (ws[k] & 0xFFFFu) ? d[8 * c + 2 * k] : 0.f;
        d[8 * c + 2 * k + 1] = (ws[k] >> 16) ? d[8 * c + 2 * k + 1] : 0.f;
    }
}

// chain layout (lane = point n, half hi) -> tile: chunk c holds features 16c + 4hi + {0..3} (slots 0..3) and 16c + 8 + 4hi + {0..3}
__device__ __forceinline__ void tile_write(uint8_t* tile, const int (&wa)[4], bf8 c0, bf8 c1) {
    const uint4 a = __builtin_bit_cast(uint4, c0), b = __builtin_bit_cast(uint4, c1);
    *reinterpret_cast<uint2*>(tile + wa[0]) = make_uint2(a.x, a.y);
    *reinterpret_cast<uint2*>(tile + wa[1]) = make_uint2(a.z, a.w);
    *reinterpret_cast<uint2*>(tile + wa[2]) = make_uint2(b.x, b.y);
    *reinterpret_cast<uint2*>(tile + wa[3]) = make_uint2(b.z, b.w);
}
// tile -> MFMA operand with the contraction over the POINTS: lane (feature l & 31, hk = l >> 5), chunk c: points 16c + 8hk + {0..7}
struct Op2 { bf8 c0, c1; };
__device__ __forceinline__ bf8 tr_chunk(const uint8_t* tile, int ra0, int ra1, int chunk) {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v)(tile + ra0 + 1024 * chunk));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v)(tile + ra1 + 1024 * chunk));
    const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8, v);
}
__device__ __forceinline__ Op2 tr_op(const uint8_t* tile, int ra0, int ra1) {
    Op2 o;
    o.c0 = tr_chunk(tile, ra0, ra1, 0);
    o.c1 = tr_chunk(tile, ra0, ra1, 1);
    return o;
}
// [rows][32 points] bf16 arrays (aux tile: 8 rows, dO tile: 4 rows): lane (row = l & (ROWS-1)), 16 bytes = its eight points;
// lanes whose feature index is >= ROWS get zeros
template <int ROWS>
__device__ __forceinline__ Op2 row_op(const uint8_t* arr, int lane) {
    const int i = lane & 31, hk = lane >> 5;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 a = *reinterpret_cast<const uint4*>(arr + (i & (ROWS - 1)) * 64 + 16 * hk);
    const uint4 b = *reinterpret_cast<const uint4*>(arr + (i & (ROWS - 1)) * 64 + 32 + 16 * hk);
    Op2 o;
    o.c0 = __builtin_bit_cast(bf8, i < ROWS ? a : z);
    o.c1 = __builtin_bit_cast(bf8, i < ROWS ? b : z);
    return o;
}
__device__ __forceinline__ bf8 onehot_col(int lane, int col) {          // B operand: column `col` all ones (bf16 1.0 = 0x3F80)
    const unsigned v = (lane & 31) == col ? 0x3F803F80u : 0u;
    return __builtin_bit_cast(bf8, make_uint4(v, v, v, v));
}

template <int I> struct IntC { static constexpr int v = I; };

#ifdef GIGA_TRACE   // diagnostic build: s_memtime timeline of workgroup 0 (tools/gpu_dect_trace.py)
static __device__ long long g_dect_trace[2 * 4 * 40];     // [occupancy head ? 1 : 0][wave][point]
#define DT(idx) do { if (BWD && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && round == 0) g_dect_trace[(id == 3) * 160 + wave * 40 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DT(idx) do {} while (0)
#endif

// NW waves of one 32-point tile each per round.  Backward: 4 (one per SIMD, 472 VGPRs).  Forward: 8 -- two per SIMD, so that one
// wave's gather runs under the other's chain -- whose C tiles sit behind the zone at a stride of CW_B bytes like the backward's.
template <bool BWD, int NW>
__global__ __launch_bounds__(NW * 64, 1) void dect_kernel(DectArgs a) {
    static_assert(!BWD || NW == 4, "the backward's tile ownership is written for four waves");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y;
    const int id = a.head_id[h];
    uint8_t* zone = smem;
    uint8_t* cw = smem + ZONE_B + wave * CW_B;                           // this wave's C tiles / aux tile / dO tile
    uint8_t* imgb = smem + LDS_FWD;
    const size_t plane_stride = (size_t)a.B * RES * RES * CD;

    // equal contiguous ranges of 32-point tiles per workgroup, an XCD's workgroups on one contiguous eighth of the points
    const long long tiles_total = (a.P + 31) / 32;
    const int bid = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const long long tile_lo = tiles_total * bid / gridDim.x, tile_hi = tiles_total * (bid + 1) / gridDim.x;
    const int rounds = (int)((tile_hi - tile_lo + NW - 1) / NW);

    dect_dma<(int)(DECT_FWD_BYTES / FRAG), NW>(a.img_fwd[h], zone, wave, lane);
    if constexpr (BWD) dect_dma<(int)(DECT_BWD_BYTES / FRAG), NW>(a.img_bwd[h], imgb, wave, lane);

    // per-lane LDS offsets, the same for every tile
    int wa[4];                                                           // chain layout -> tile pieces (tile_write)
#pragma unroll
    for (int k = 0; k < 4; ++k) wa[k] = piece_addr(n, 4 * (k >> 1) + 2 * (k & 1) + hi);
    int ca[4];                                                           // C tile -> chain B operand: chunk half hf, pieces 4hf + 2hi, + 1
#pragma unroll
    for (int k = 0; k < 4; ++k) ca[k] = piece_addr(n, 4 * (k >> 1) + 2 * hi + (k & 1));
    const int rp = 8 * (lane >> 5) + ((lane & 15) >> 2), rq = 4 * ((lane >> 4) & 1) + (lane & 3);
    const int ra0 = piece_addr(rp, rq), ra1 = piece_addr(rp + 4, rq);    // transposing reads: points +0..3 and +4..7 of the lane's eight

    f32x16 acc[DECT_ACC];
    if constexpr (BWD) {
#pragma unroll
        for (int s = 0; s < DECT_ACC; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    }

    for (int round = 0; round < rounds; ++round) {
        const long long tile = tile_lo + (long long)round * NW + wave;
        const bool active = tile < tile_hi;
        long long g = tile * 32 + n;
        const bool valid = active && g < a.P;
        if (!valid) g = a.P - 1;
        DT(0);
        // ---------------- gather: four adjacent lanes fetch the four 16-byte quads of one (point, channel half) = one 64-byte line,
        // interpolate (the fma order of aten's grid_sampler: nw, ne, sw, se), round to bf16 and write the 8-byte piece of the C
        // tile; the chain then reads its B operands (lane = point) from the tile.  The tile is also the X operand of fc_c's
        // weight gradient.
        const long long tile0 = tile * 32;
        {
            // phase 1: the coordinates of this lane's four (point, piece) items; phase 2: 2 x 24 tap loads (two round trips instead of
            // one per item: left to itself the compiler waits for every item's coordinates before it asks for the item's taps);
            // phase 3: interpolate, round, store.  The sched_barriers pin the phases.
            float cx[4], cy[4], cz[4];
            int bp[4];
            const int q = lane & 7;                                      // piece = channels 4q .. 4q+3
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const int pt = part * 8 + (lane >> 3);
                long long gp = active ? tile0 + pt : a.P - 1;
                if (gp >= a.P) gp = a.P - 1;
                int rdummy;
                split_scene(gp, a.N, a.invN, bp[part], rdummy);
                cx[part] = a.p[3 * gp + 0]; cy[part] = a.p[3 * gp + 1]; cz[part] = a.p[3 * gp + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int half = 0; half < 2; ++half) {                       // 24 loads (96 VGPRs) in flight per half
                float4 v[2][3][4];
                float wq[2][3][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int part = 2 * half + u;
                    const float nx = norm_coord(cx[part]), ny = norm_coord(cy[part]), nz = norm_coord(cz[part]);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const Bilin bl = bilin_setup(pl == 2 ? ny : nx, pl == 1 ? ny : nz);   // xz: (x, z)  xy: (x, y)  yz: (y, z)
                        const float* base = a.planes + pl * plane_stride + (size_t)bp[part] * RES * RES * CD + 4 * q;
                        v[u][pl][0] = *reinterpret_cast<const float4*>(base + (size_t)bl.o00 * CD);
                        v[u][pl][1] = *reinterpret_cast<const float4*>(base + (size_t)bl.o01 * CD);
                        v[u][pl][2] = *reinterpret_cast<const float4*>(base + (size_t)bl.o10 * CD);
                        v[u][pl][3] = *reinterpret_cast<const float4*>(base + (size_t)bl.o11 * CD);
                        wq[u][pl][0] = bl.w00; wq[u][pl][1] = bl.w01; wq[u][pl][2] = bl.w10; wq[u][pl][3] = bl.w11;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // (the interpolation below is pure arithmetic, which the optimiser would hoist back in front of the barrier and interleave
                //  with the loads, a handful in flight at a time: its weights pass through an opaque asm that stays behind the barrier)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        asm volatile("" : "+v"(wq[u][pl][0]), "+v"(wq[u][pl][1]), "+v"(wq[u][pl][2]), "+v"(wq[u][pl][3]));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int pt = (2 * half + u) * 8 + (lane >> 3);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const float4 v00 = v[u][pl][0], v01 = v[u][pl][1], v10 = v[u][pl][2], v11 = v[u][pl][3];
                        const float w00 = wq[u][pl][0], w01 = wq[u][pl][1], w10 = wq[u][pl][2], w11 = wq[u][pl][3];
                        bf4 o4;
                        o4[0] = (__bf16)fmaf(v11.x, w11, fmaf(v10.x, w10, fmaf(v01.x, w01, v00.x * w00)));
                        o4[1] = (__bf16)fmaf(v11.y, w11, fmaf(v10.y, w10, fmaf(v01.y, w01, v00.y * w00)));
                        o4[2] = (__bf16)fmaf(v11.z, w11, fmaf(v10.z, w10, fmaf(v01.z, w01, v00.z * w00)));
                        o4[3] = (__bf16)fmaf(v11.w, w11, fmaf(v10.w, w10, fmaf(v01.w, w01, v00.w * w00)));
                        *reinterpret_cast<bf4*>(cw + pl * TILE_B + piece_addr(pt, q)) = o4;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // aux operand of the chain: [p_hi (3), 1, p_lo (3), 1] on hi = 0 lanes, [p_hi (3), 0 ...] on hi = 1 lanes (giga_dect.h)
        const float px = a.p[3 * g + 0], py = a.p[3 * g + 1], pz = a.p[3 * g + 2];
        const __bf16 xh = (__bf16)px, yh = (__bf16)py, zh = (__bf16)pz;
        const __bf16 xl = (__bf16)(px - (float)xh), yl = (__bf16)(py - (float)yh), zl = (__bf16)(pz - (float)zh);
        const __bf16 one = (__bf16)1.0f, zero = (__bf16)0.0f;
        bf8 av = {xh, yh, zh, zero, zero, zero, zero, zero};
        if (hi == 0) { av[3] = one; av[4] = xl; av[5] = yl; av[6] = zl; av[7] = one; }
        if constexpr (BWD) {
            // aux tile for the weight gradients, [column][point]: p_hi (0..2), 1 (3), p_lo (4..6), 0 (7)
            if (hi == 0) {
                __bf16* at = reinterpret_cast<__bf16*>(cw + 3 * TILE_B);
                at[0 * 32 + n] = xh; at[1 * 32 + n] = yh; at[2 * 32 + n] = zh; at[3 * 32 + n] = one;
                at[4 * 32 + n] = xl; at[5 * 32 + n] = yl; at[6 * 32 + n] = zl; at[7 * 32 + n] = zero;
            }
        }
        DT(1);
        // the forward image (first round: requested at kernel start; later rounds: after the previous round's last barrier) has landed
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        DT(2);
        bf8 cfb[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint2 lo = *reinterpret_cast<const uint2*>(cw + (c >> 1) * TILE_B + ca[2 * (c & 1)]);
            const uint2 hi2 = *reinterpret_cast<const uint2*>(cw + (c >> 1) * TILE_B + ca[2 * (c & 1) + 1]);
            cfb[c] = __builtin_bit_cast(bf8, make_uint4(lo.x, lo.y, hi2.x, hi2.y));
        }
        // ---------------- forward chain (fragment order of the image: giga_dect.h) ----------------------------------------
        const bf8* W = reinterpret_cast<const bf8*>(zone);
        const float* ctab = reinterpret_cast<const float*>(zone + (size_t)DECT_FWD_FRAGS * FRAG);
        bf8 XN[NBLK][2], XH[NBLK][2], XO[2];
        f32x16 net, o;
#pragma unroll
        for (int r = 0; r < 16; ++r) net[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 7; ++c) net = mfma_bf(W[c * 64 + lane], c < 6 ? cfb[c] : av, net);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int k = 11 * blk;
            f32x16 hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ctab + blk * 32 + 8 * q + 4 * hi);
                hh[4 * q + 0] = v.x; hh[4 * q + 1] = v.y; hh[4 * q + 2] = v.z; hh[4 * q + 3] = v.w;
            }
            XN[blk][0] = relu_bf8(net, 0); XN[blk][1] = relu_bf8(net, 1);
            hh = mfma_bf(W[(k + 7) * 64 + lane], XN[blk][0], hh);
            hh = mfma_bf(W[(k + 8) * 64 + lane], XN[blk][1], hh);
            if (blk + 1 < NBLK) {            // the next block's fc_c (+ folded biases) accumulates between fc_0 and fc_1
#pragma unroll
                for (int c = 0; c < 7; ++c) net = mfma_bf(W[(k + 11 + c) * 64 + lane], c < 6 ? cfb[c] : av, net);
            } else {
                net = mfma_bf(W[55 * 64 + lane], av, net);               // + bias of the last fc_1
            }
            XH[blk][0] = relu_bf8(hh, 0); XH[blk][1] = relu_bf8(hh, 1);
            net = mfma_bf(W[(k + 9) * 64 + lane], XH[blk][0], net);
            net = mfma_bf(W[(k + 10) * 64 + lane], XH[blk][1], net);
        }
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ctab + NBLK * 32 + 8 * q + 4 * hi);
                o[4 * q + 0] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
            }
            XO[0] = relu_bf8(net, 0); XO[1] = relu_bf8(net, 1);
            o = mfma_bf(W[56 * 64 + lane], XO[0], o);
            o = mfma_bf(W[57 * 64 + lane], XO[1], o);
        }
        DT(3);
        if constexpr (!BWD) {
            if (hi == 0 && valid) {
                float d0 = o[0], d1 = o[1], d2 = o[2], d3 = o[3];
                float* dst = a.out[h];
                if (id == 1) {
                    if (a.post) {                                        // F.normalize(dim=2): x / max(||x||_2, 1e-12)
                        const float inv = 1.0f / fmaxf(sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3), 1e-12f);
                        d0 *= inv; d1 *= inv; d2 *= inv; d3 *= inv;
                    }
                    *reinterpret_cast<float4*>(dst + 4 * g) = make_float4(d0, d1, d2, d3);
                } else {
                    if (id == 0 && a.post) d0 = 1.0f / (1.0f + expf(-d0));
                    dst[g] = d0;
                }
            }
            continue;                                                    // (the image stays; a wave's C tiles are its own)
        }
        if constexpr (BWD) {
            // ---------------- epilogue backward: dO (<= 4 values per point), in both lane halves -------------------------------
            float dO[4] = {0.f, 0.f, 0.f, 0.f};
            if (id == 1) {               // rot = z / max(|z|, eps): dz = (dr - r (r . dr)) / max(|z|, eps)   (F.normalize backward)
                float z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = __shfl(o[k], n);      // the raw outputs live in the hi = 0 lanes (rows 0..3)
                const float4 r4 = *reinterpret_cast<const float4*>(a.out[h] + 4 * g);
                const float4 d4 = *reinterpret_cast<const float4*>(a.dout[h] + 4 * g);
                const float inv = 1.0f / fmaxf(sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2] + z[3] * z[3]), 1e-12f);
                const float rd = r4.x * d4.x + r4.y * d4.y + r4.z * d4.z + r4.w * d4.w;
                dO[0] = (d4.x - r4.x * rd) * inv; dO[1] = (d4.y - r4.y * rd) * inv;
                dO[2] = (d4.z - r4.z * rd) * inv; dO[3] = (d4.w - r4.w * rd) * inv;
            } else {
                float d = a.dout[h][g];
                if (id == 0) { const float qv = a.out[h][g]; d *= qv * (1.0f - qv); }       // sigmoid'
                dO[0] = d;
            }
            if (!valid) { dO[0] = 0.f; dO[1] = 0.f; dO[2] = 0.f; dO[3] = 0.f; }
            __syncthreads();             // every wave is through its forward chain: the zone becomes the waves' step buffers
            DT(4);
            uint8_t* zw = zone + wave * ZONE_WAVE;
            // ---------------- step "5": XO, dO, DN[5] = (Wout^T dO) * (net5 > 0) ------------------------------------------------
            const float* wout = reinterpret_cast<const float*>(imgb + (size_t)DECT_BWD_FRAGS * FRAG);     // [4][32], bf16 values
            if (hi == 0) {
                __bf16* dt = reinterpret_cast<__bf16*>(cw + 3 * TILE_B + 512);
#pragma unroll
                for (int k = 0; k < 4; ++k) dt[k * 32 + n] = (__bf16)dO[k];
            }
            tile_write(zw + 3 * TILE_B + 2 * TILE_B, wa, XO[0], XO[1]);                    // buffer 1, slot XN
            f32x16 G;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = (r & 3) + 8 * (r >> 2) + 4 * hi;
                G[r] = wout[f] * dO[0] + wout[32 + f] * dO[1] + wout[64 + f] * dO[2] + wout[96 + f] * dO[3];
            }
            mask_by(G, 0, XO[0]); mask_by(G, 1, XO[1]);
            bf8 Gb[2] = {cvt_bf8(G, 0), cvt_bf8(G, 1)};
            tile_write(zw + 6 * TILE_B + (5 % 3) * TILE_B, wa, Gb[0], Gb[1]);
            DT(5);
            __syncthreads();
            DT(6);
            if (wave == 2) {             // fc_out: dW[o][k] = sum_p dO[p][o] XO[p][k]
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    const Op2 A = row_op<4>(smem + ZONE_B + w2 * CW_B + 3 * TILE_B + 512, lane);
                    const Op2 B = tr_op(zone + w2 * ZONE_WAVE + 5 * TILE_B, ra0, ra1);
                    acc[7] = mfma_bf(A.c0, B.c0, acc[7]);
                    acc[7] = mfma_bf(A.c1, B.c1, acc[7]);
                }
            }
            f32x16 dc[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int r = 0; r < 16; ++r) dc[pl][r] = 0.f;
            const bf8* WB = reinterpret_cast<const bf8*>(imgb);
            auto step = [&](auto ic) __attribute__((always_inline)) {
                constexpr int blk = decltype(ic)::v;
                constexpr int kb = 10 * blk;
                uint8_t* buf = zw + (blk & 1) * 3 * TILE_B;
                f32x16 dh, dn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { dh[r] = 0.f; dn[r] = 0.f; }
                dh = mfma_bf(WB[(kb + 8) * 64 + lane], Gb[0], dh);       // W1^T G
                dh = mfma_bf(WB[(kb + 9) * 64 + lane], Gb[1], dh);
                mask_by(dh, 0, XH[blk][0]); mask_by(dh, 1, XH[blk][1]);
                const bf8 Hb0 = cvt_bf8(dh, 0), Hb1 = cvt_bf8(dh, 1);
                dn = mfma_bf(WB[(kb + 6) * 64 + lane], Hb0, dn);         // W0^T DH
                dn = mfma_bf(WB[(kb + 7) * 64 + lane], Hb1, dn);
                mask_by(dn, 0, XN[blk][0]); mask_by(dn, 1, XN[blk][1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) G[r] += dn[r];              // DN[blk]
                Gb[0] = cvt_bf8(G, 0); Gb[1] = cvt_bf8(G, 1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {                         // dc += Wc^T DN[blk]
                    dc[pl] = mfma_bf(WB[(kb + 2 * pl) * 64 + lane], Gb[0], dc[pl]);
                    dc[pl] = mfma_bf(WB[(kb + 2 * pl + 1) * 64 + lane], Gb[1], dc[pl]);
                }
                tile_write(buf, wa, XH[blk][0], XH[blk][1]);
                tile_write(buf + TILE_B, wa, Hb0, Hb1);
                tile_write(buf + 2 * TILE_B, wa, XN[blk][0], XN[blk][1]);
                tile_write(zw + 6 * TILE_B + (blk % 3) * TILE_B, wa, Gb[0], Gb[1]);
                if (blk == 0 && a.dcbuf && valid) {                      // rows of [P][96] for plane_gather_kernel
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(a.dcbuf + g * 96 + pl * 32 + 8 * q + 4 * hi) =
                                make_float4(dc[pl][4 * q], dc[pl][4 * q + 1], dc[pl][4 * q + 2], dc[pl][4 * q + 3]);
                }
                DT(7 + 3 * (4 - blk));
                __syncthreads();
                DT(8 + 3 * (4 - blk));
                // ---- weight-gradient tiles of this block: wave w plays role (w + blk) & 3 -------------------------------------
                auto role = [&](auto wc) __attribute__((always_inline)) {
                    constexpr int WV = decltype(wc)::v;
                    constexpr int R = (WV + blk) & 3, S = dect_slot(WV, blk);
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const uint8_t* z2 = zone + w2 * ZONE_WAVE;
                        const uint8_t* b2 = z2 + (blk & 1) * 3 * TILE_B;
                        const uint8_t* c2 = smem + ZONE_B + w2 * CW_B;
                        if constexpr (R == 0) {                           // fc_c, input planes 0 and 1
                            const Op2 A = tr_op(z2 + 6 * TILE_B + (blk % 3) * TILE_B, ra0, ra1);
                            const Op2 B0 = tr_op(c2, ra0, ra1), B1 = tr_op(c2 + TILE_B, ra0, ra1);
                            acc[S] = mfma_bf(A.c0, B0.c0, acc[S]); acc[S] = mfma_bf(A.c1, B0.c1, acc[S]);
                            acc[S + 1] = mfma_bf(A.c0, B1.c0, acc[S + 1]); acc[S + 1] = mfma_bf(A.c1, B1.c1, acc[S + 1]);
                        } else if constexpr (R == 1) {                    // fc_c plane 2; bias tile
                            const Op2 A = tr_op(z2 + 6 * TILE_B + (blk % 3) * TILE_B, ra0, ra1);
                            const Op2 B2 = tr_op(c2 + 2 * TILE_B, ra0, ra1);
                            acc[S] = mfma_bf(A.c0, B2.c0, acc[S]); acc[S] = mfma_bf(A.c1, B2.c1, acc[S]);
                            const Op2 X = row_op<8>(c2 + 3 * TILE_B, lane);
                            acc[S + 1] = mfma_bf(A.c0, X.c0, acc[S + 1]); acc[S + 1] = mfma_bf(A.c1, X.c1, acc[S + 1]);
                            const Op2 H = tr_op(b2 + TILE_B, ra0, ra1);
                            const bf8 one8 = onehot_col(lane, 8);
                            acc[S + 1] = mfma_bf(H.c0, one8, acc[S + 1]); acc[S + 1] = mfma_bf(H.c1, one8, acc[S + 1]);
                            if constexpr (blk == NBLK - 1) {
                                const Op2 A5 = tr_op(z2 + 6 * TILE_B + (5 % 3) * TILE_B, ra0, ra1);
                                const bf8 one9 = onehot_col(lane, 9), one10 = onehot_col(lane, 10);
                                acc[S + 1] = mfma_bf(A5.c0, one9, acc[S + 1]); acc[S + 1] = mfma_bf(A5.c1, one9, acc[S + 1]);
                                const Op2 D = row_op<4>(c2 + 3 * TILE_B + 512, lane);
                                acc[S + 1] = mfma_bf(D.c0, one10, acc[S + 1]); acc[S + 1] = mfma_bf(D.c1, one10, acc[S + 1]);
                            }
                        } else if constexpr (R == 2) {                    // fc_1: DN[blk + 1]^T XH[blk]
                            const Op2 A = tr_op(z2 + 6 * TILE_B + ((blk + 1) % 3) * TILE_B, ra0, ra1);
                            const Op2 B = tr_op(b2, ra0, ra1);
                            acc[S] = mfma_bf(A.c0, B.c0, acc[S]); acc[S] = mfma_bf(A.c1, B.c1, acc[S]);
                        } else {                                          // fc_0: DH[blk]^T XN[blk]
                            const Op2 A = tr_op(b2 + TILE_B, ra0, ra1);
                            const Op2 B = tr_op(b2 + 2 * TILE_B, ra0, ra1);
                            acc[S] = mfma_bf(A.c0, B.c0, acc[S]); acc[S] = mfma_bf(A.c1, B.c1, acc[S]);
                        }
                    }
                };
                if (wave == 0) role(IntC<0>{});
                else if (wave == 1) role(IntC<1>{});
                else if (wave == 2) role(IntC<2>{});
                else role(IntC<3>{});
                DT(9 + 3 * (4 - blk));
            };
            step(IntC<4>{}); step(IntC<3>{}); step(IntC<2>{}); step(IntC<1>{}); step(IntC<0>{});
            __syncthreads();             // every wave has read what it needs of this round's tiles
            DT(22);
            if (a.gplanes && !a.dcbuf) {
                // ---------------- scatter dc into the plane gradients (sample_plane_feature backward) with fp32 atomics, transposed
                // through a wave-private stage in the zone so that one atomic instruction covers two (point, tap) pairs x 32
                // contiguous channels (giga_decoder_bwd.hip)
                float* T = reinterpret_cast<float*>(zw);                  // [point][96] values
                int* Q = reinterpret_cast<int*>(T + 32 * 96);             // [point][plane][tap] offsets
                float* Wt = reinterpret_cast<float*>(Q + 32 * 12);        // [point][plane][tap] weights
                int bsc, rdummy;
                split_scene(g, a.N, a.invN, bsc, rdummy);
                const float nx = norm_coord(px), ny = norm_coord(py), nz = norm_coord(pz);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(T + n * 96 + pl * 32 + 8 * q + 4 * hi) =
                            make_float4(dc[pl][4 * q], dc[pl][4 * q + 1], dc[pl][4 * q + 2], dc[pl][4 * q + 3]);
                    if (hi == 0) {
                        const Bilin bl = bilin_setup(pl == 2 ? ny : nx, pl == 1 ? ny : nz);
                        const int base = (int)(pl * plane_stride + (size_t)bsc * RES * RES * CD);
                        const int o4[4] = {bl.o00, bl.o01, bl.o10, bl.o11};
                        const float w4[4] = {bl.w00, bl.w01, bl.w10, bl.w11};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            Q[(n * 3 + pl) * 4 + t] = valid ? base + o4[t] * CD : -1;
                            Wt[(n * 3 + pl) * 4 + t] = w4[t];
                        }
                    }
                }
                // every wave's (point pair, plane) items are dealt over the FOUR waves: float atomics retire slowly (~180 clocks per wave
                // instruction), and with one query per scene a single wave of the workgroup has points at all
                __syncthreads();
                const int c = lane & 31;
                for (int item = wave; item < 4 * 48; item += 4) {
                    const int w2 = item / 48, rest = item - 48 * w2, pl = rest >> 4, pr = rest & 15;
                    if ((tile_lo + (long long)round * NW + w2) >= tile_hi) continue;      // that wave had no tile (uniform)
                    const float* T2 = reinterpret_cast<const float*>(zone + w2 * ZONE_WAVE);
                    const int* Q2 = reinterpret_cast<const int*>(T2 + 32 * 96);
                    const float* W2 = reinterpret_cast<const float*>(Q2 + 32 * 12);
                    const int pt = 2 * pr + hi;
                    const float v = T2[pt * 96 + pl * 32 + c];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int off = Q2[(pt * 3 + pl) * 4 + t];
                        const float w = W2[(pt * 3 + pl) * 4 + t];
                        if (off >= 0) atomicAdd(a.gplanes + off + c, v * w);
                    }
                }
                __syncthreads();         // the stage is free before the next round's forward image lands on it
            }
            DT(23);
            if (round + 1 < rounds) dect_dma<(int)(DECT_FWD_BYTES / FRAG), NW>(a.img_fwd[h], zone, wave, lane);
        }
    }
    if constexpr (BWD) {
        // ---------------- the workgroup's partial gradient tiles, D-register layout (tile, register, lane): coalesced 256-byte stores
        float* part = a.partial[h] + (size_t)blockIdx.x * DECT_NTILES * DECT_TILE_FLOATS;
        auto store_tiles = [&](auto wc) __attribute__((always_inline)) {
            constexpr int WV = decltype(wc)::v;
            auto one_block = [&](auto bc) __attribute__((always_inline)) {
                constexpr int blk = decltype(bc)::v;
                constexpr int R = (WV + blk) & 3, S = dect_slot(WV, blk), t0 = dect_role_tile(R), NT = dect_role_ntiles(R);
#pragma unroll
                for (int r = 0; r < 16; ++r) part[(size_t)(6 * blk + t0) * DECT_TILE_FLOATS + r * 64 + lane] = acc[S][r];
                if constexpr (NT == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[(size_t)(6 * blk + t0 + 1) * DECT_TILE_FLOATS + r * 64 + lane] = acc[S + 1][r];
                }
            };
            one_block(IntC<0>{}); one_block(IntC<1>{}); one_block(IntC<2>{}); one_block(IntC<3>{}); one_block(IntC<4>{});
            if constexpr (WV == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) part[(size_t)30 * DECT_TILE_FLOATS + r * 64 + lane] = acc[7][r];
            }
        };
        if (wave == 0) store_tiles(IntC<0>{});
        else if (wave == 1) store_tiles(IntC<1>{});
        else if (wave == 2) store_tiles(IntC<2>{});
        else store_tiles(IntC<3>{});
    }
}

// ------------------------------- partial tiles -> flat gradient buffer --------------------------------------------------------
// One workgroup per (tile, quarter of its 1024 elements, head): four waves sum interleaved quarters of the partial images (eight
// loads in flight), fold through LDS in a fixed order and write.  Weight tiles are plain stores; the few bias / fc_p values of a bias
// tile are ADDED (two columns fold into one fc_p element, one column feeds two biases): the caller has zeroed the gradient buffer.
struct DectReduceArgs {
    const float* partial[NHEADS];
    int nwg[NHEADS];
    int head_id[NHEADS];
    HeadParamOff off[NHEADS];
    float* grads;
    int nheads;
};
__global__ __launch_bounds__(256) void dect_reduce_kernel(DectReduceArgs a) {
    __shared__ float4 fold[3][64];
    const int tile = blockIdx.x >> 2, quarter = blockIdx.x & 3, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* src = a.partial[h] + (size_t)tile * DECT_TILE_FLOATS + quarter * 256 + lane * 4;
    const int nwg = a.nwg[h];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w0 = wave; w0 < nwg; w0 += 32) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w0 + 4 * u;
            v[u] = w < nwg ? *reinterpret_cast<const float4*>(src + (size_t)w * DECT_NTILES * DECT_TILE_FLOATS) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    if (wave > 0) fold[wave - 1][lane] = s;
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float4 t = fold[k][lane]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    const float vals[4] = {s.x, s.y, s.z, s.w};
    const HeadParamOff& o = a.off[h];
    const int out_dim = HEAD_OUT[a.head_id[h]];
    float* G = a.grads;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = quarter * 256 + lane * 4 + e;          // = r * 64 + lane' of the D layout
        const int r = idx >> 6, l2 = idx & 63;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l2 >> 5), col = l2 & 31;      // dW[row = output feature][col = input feature]
        const float v = vals[e];
        if (tile == 30) {
            if (row < out_dim) G[o.out_w + row * 32 + col] = v;
            continue;
        }
        const int blk = tile / 6, t = tile % 6;
        if (t < 3) G[o.fc_c_w[blk] + row * 96 + 32 * t + col] = v;
        else if (t == 4) G[o.fc1_w[blk] + row * 32 + col] = v;
        else if (t == 5) G[o.fc0_w[blk] + row * 32 + col] = v;
        else {                                                  // bias tile
            if (col == 3) {
                atomicAdd(G + o.fc_c_b[blk] + row, v);
                if (blk > 0) atomicAdd(G + o.fc1_b[blk - 1] + row, v);
                else atomicAdd(G + o.fc_p_b + row, v);
            } else if (col < 7) {
                if (blk == 0) atomicAdd(G + o.fc_p_w + row * 3 + (col & 3), v);
            } else if (col == 8) atomicAdd(G + o.fc0_b[blk] + row, v);
            else if (col == 9) { if (blk == NBLK - 1) atomicAdd(G + o.fc1_b[NBLK - 1] + row, v); }
            else if (col == 10) { if (blk == NBLK - 1 && row < out_dim) atomicAdd(G + o.out_b + row, v); }
        }
    }
}

// ------------------------------- launchers ---------------------------------------------------------------------------------------
static int dect_grid(long long P) {
    const long long tiles = (P + 31) / 32;
    const long long wgs = (tiles + 3) / 4;
    return (int)(wgs < 256 ? wgs : 256);
}

int launch_dect_forward(const float* planes, const float* p, const uint8_t* blob, int head_mask, float* const* outs, int B, int N,
                        int post, hipStream_t s) {
    const long long P = (long long)B * N;
    if (P <= 0 || (head_mask & 15) == 0) return 0;
    const PackOff ko = pack_offsets();
    DectArgs a{};
    a.planes = planes; a.p = p; a.B = B; a.N = N; a.P = P; a.invN = 1.0f / (float)N; a.post = post;
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_mask >> h & 1)) continue;
        a.head_id[a.nheads] = h; a.img_fwd[a.nheads] = blob + ko.dect[h]; a.out[a.nheads] = outs[h];
        ++a.nheads;
    }
    const long long tiles = (P + 31) / 32, wgs = (tiles + FWD_NW - 1) / FWD_NW;
    giga::dyn_lds_once(reinterpret_cast<const void*>(dect_kernel<false, FWD_NW>), LDS_FWD8);
    GIGA_LAUNCH((dect_kernel<false, FWD_NW>), dim3((unsigned)(wgs < 256 ? wgs : 256), a.nheads), dim3(FWD_NW * 64), LDS_FWD8, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

// scratch of one backward call: the partial tiles of its heads (+ the dc rows when the plane gradient is gathered)
size_t dect_partial_floats(long long P, int nheads) { return P > 0 ? (size_t)dect_grid(P) * DECT_NTILES * DECT_TILE_FLOATS * nheads : 0; }

// One decoder-backward call on the bf16 kernels.  `scratch` receives the partial tiles (dect_partial_floats) and, with dcrows, the
// [P][96] rows behind them; the reduce is enqueued by launch_dect_reduce once all calls of the step have run.
int launch_dect_backward(const float* planes, const float* p, const uint8_t* blob, const uint8_t* bwd_blob, int head_mask,
                         const float* const* outs, const float* const* douts, float* gplanes, float* dcbuf, float* scratch, int B,
                         int N, hipStream_t s, DectPending* pend) {
    const long long P = (long long)B * N;
    if (P <= 0 || (head_mask & 15) == 0) return 0;
    const PackOff ko = pack_offsets();
    const BwdPackOff bo = bwd_pack_offsets();
    DectArgs a{};
    a.planes = planes; a.p = p; a.B = B; a.N = N; a.P = P; a.invN = 1.0f / (float)N;
    a.gplanes = gplanes; a.dcbuf = dcbuf;
    const int grid = dect_grid(P);
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_mask >> h & 1)) continue;
        a.head_id[a.nheads] = h; a.img_fwd[a.nheads] = blob + ko.dect[h]; a.img_bwd[a.nheads] = bwd_blob + bo.dect[h];
        a.out[a.nheads] = const_cast<float*>(outs[h]); a.dout[a.nheads] = douts[h];
        a.partial[a.nheads] = scratch + (size_t)a.nheads * grid * DECT_NTILES * DECT_TILE_FLOATS;
        pend->partial[pend->n] = a.partial[a.nheads]; pend->nwg[pend->n] = grid; pend->head_id[pend->n] = h; ++pend->n;
        ++a.nheads;
    }
    giga::dyn_lds_once(reinterpret_cast<const void*>(dect_kernel<true, 4>), LDS_BWD);
    GIGA_LAUNCH((dect_kernel<true, 4>), dim3(grid, a.nheads), dim3(256), LDS_BWD, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_dect_reduce(const DectPending& pend, float* grads, int head_present, hipStream_t s) {
    if (pend.n == 0) return 0;
    const ParamOff po = param_offsets(head_present);
    DectReduceArgs r{};
    r.grads = grads; r.nheads = pend.n;
    for (int k = 0; k < pend.n; ++k) {
        r.partial[k] = pend.partial[k]; r.nwg[k] = pend.nwg[k]; r.head_id[k] = pend.head_id[k]; r.off[k] = po.head[pend.head_id[k]];
    }
    GIGA_LAUNCH(dect_reduce_kernel, dim3(DECT_NTILES * 4, pend.n), dim3(256), 0, s, r);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
