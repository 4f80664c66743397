// conv32: the weight-stationary, LDS-image-resident convolution of the f16-class U-Net (f16, f16x3 split and bf16 arithmetic).
// Geometry, the reasons for it and the reference citations: giga_conv32_geom.h.  This file is the gfx950 side:
//   c32_load_weights : a wave pulls the fragments of its slice group straight from the packed blob (L2) into REGISTERS -- issued
//                      before the group barrier of the persistent kernel, so they land while the barrier is waited for;
//   c32_stage        : the workgroup copies its haloed sub-band into LDS (16-byte vectors, pad columns / rows written as zeros;
//                      2x2 max-pool, f16x3 split or bf16 rounding applied on the way);
//   c32_mma          : per (tile, tap, k-chunk) ONE ds_read_b128 per lane at `lane base + immediate` feeds SPW MFMAs whose A
//                      operands never leave the register file; a ring of four reads is in flight per tile;
//   epilogue         : a lane holds 16 consecutive output channels of one pixel: bias (accumulator init), ReLU, rounding, one or
//                      two 16-byte stores per 8 channels.
// The summation order of every output is fixed (bias, then taps x k-chunks in order), so results do not depend on how the rows
// are dealt out: the persistent kernel, the per-layer launches and every batch size agree bit for bit.
#pragma once
#include <type_traits>

#include "giga_conv16.h"
#include "giga_conv32_geom.h"

namespace giga {

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma16_bf(bf16x8v a, bf16x8v b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#ifdef GIGA_TRACE   // diagnostic build: s_memtime stamps of member 0 of the group that holds image 0, per layer and wave
static __device__ long long g_c32_trace[NCONV][C32_NW][8];
#define C32_T(a, idx) do { if ((a).trace_id >= 0 && member == 0 && (threadIdx.x & 63) == 0) \
        g_c32_trace[(a).trace_id][threadIdx.x >> 6][idx] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C32_T(a, idx) do {} while (0)
#endif

template <class G>
struct C32W { uint4 v[G::SPW][G::TAPS * G::KCP][G::NOP]; };

// fragments of slice group sg = wave % SG, k-part `part`: [slice][tap][k-chunk] x NOP, 16 bytes per lane each
template <class G>
__device__ __forceinline__ void c32_load_weights(const ConvArgs& a, C32W<G>& w, int part) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sg = wave % G::SG;
    const uint8_t* base = a.w + lane * 16;
#pragma unroll
    for (int s = 0; s < G::SPW; ++s)
#pragma unroll
        for (int tap = 0; tap < G::TAPS; ++tap)
#pragma unroll
            for (int kcp = 0; kcp < G::KCP; ++kcp) {
                const int f = G::frag(sg * G::SPW + s, tap, part * G::KCP + kcp);
#pragma unroll
                for (int o = 0; o < G::NOP; ++o)
                    w.v[s][tap * G::KCP + kcp][o] = *reinterpret_cast<const uint4*>(base + (size_t)(f * G::NOP + o) * FRAG);
            }
}

// ---- staging: the haloed sub-band [sb - HALO, sb + R + HALO) x P pixels -> LDS ------------------------------------------------
// All of a thread's loads (up to U items) are in flight before the first LDS write; the zero pixels are written under their
// latency.  Thread t takes items t, t + 256, ... of the sub-band's real rows (giga_conv32_geom.h: Cur).
template <class G>
__device__ __forceinline__ void c32_stage(const ConvArgs& a, uint8_t* smem, int sb, int R) {
    constexpr int MODE = G::MODE, ES = G::ES;
    constexpr int NPOS = G::POOLIN ? 4 : 1;                  // source pixels per staged pixel
    constexpr int VPI = MODE == C32_NATIVE ? 1 : 2;          // 16-byte source vectors per item (8 channels)
    constexpr int U = (MODE == C32_NATIVE ? 24 : 12) / NPOS; // items per thread in flight
    const int tid = threadIdx.x, Gimg = a.nimg;
    int rrA, rrB;
    G::real_rows(sb, R, Gimg, rrA, rrB);
    const int nA = (rrB - rrA) * G::RI;
    const char* in0 = reinterpret_cast<const char*>(a.in0);
    const char* in1 = reinterpret_cast<const char*>(a.in1);
    typename G::Cur k = G::cur_init(rrA, tid);
    bool first_chunk = true;
    for (int j0 = tid;; j0 += G::NTHR * U) {
        uint4 v[U][NPOS][VPI];
        int lds[U];
        uint32_t qoff[U];
        bool ok[U], own[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = j0 + G::NTHR * u < nA;
            lds[u] = G::cur_lds(k, sb);
            const int ch0 = G::cur_ch(k);
            const bool first = G::C1 == 0 || ch0 < G::C0;
            const char* src = first ? in0 : in1;
            const uint32_t C = first ? G::C0 : G::C1, ch = first ? ch0 : ch0 - G::C0;
#pragma unroll
            for (int q = 0; q < NPOS; ++q) {
                const uint32_t off = ok[u] ? ((uint32_t)G::cur_src_pixel(k, q) * C + ch) * (uint32_t)ES : 0u;   // (clamped: every load is issued)
#pragma unroll
                for (int e = 0; e < VPI; ++e) v[u][q][e] = *reinterpret_cast<const uint4*>(src + off + 16 * e);
            }
            if constexpr (G::POOLIN) {
                own[u] = ok[u] && G::cur_own(k, sb, R);
                qoff[u] = ((uint32_t)(k.rr * G::W + G::cur_x(k)) * G::C0 + ch0) * (uint32_t)ES;
            }
            G::cur_next(k);
        }
        if (first_chunk) {                                   // zero rows / zero columns, under the latency of the loads
            first_chunk = false;
            if constexpr (G::HALO) {
                const int nq = G::n_buf_pixels(R);
                for (int q = tid; q < nq; q += G::NTHR)
                    if (G::pad_pixel(q, sb, Gimg)) {
#pragma unroll
                        for (int e = 0; e < G::IPP * G::ILB / 16; ++e)
                            *reinterpret_cast<uint4*>(smem + q * G::PS + 16 * e) = make_uint4(0, 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            uint8_t* dst = smem + lds[u];
            if constexpr (MODE == C32_NATIVE) {
                half8 x = __builtin_bit_cast(half8, v[u][0][0]);
#pragma unroll
                for (int q = 1; q < NPOS; ++q) x = __builtin_elementwise_max(x, __builtin_bit_cast(half8, v[u][q][0]));
                *reinterpret_cast<half8*>(dst) = x;
                if constexpr (G::POOLIN) {
                    if (own[u] && a.out_pool) *reinterpret_cast<half8*>(reinterpret_cast<char*>(a.out_pool) + qoff[u]) = x;
                }
            } else {
                float x[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    f32x4v f = __builtin_bit_cast(f32x4v, v[u][0][e2]);
#pragma unroll
                    for (int q = 1; q < NPOS; ++q) {
                        const f32x4v g = __builtin_bit_cast(f32x4v, v[u][q][e2]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = __builtin_fmaxf(f[e], g[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * e2 + e] = f[e];
                }
                if constexpr (G::POOLIN) {
                    if (own[u] && a.out_pool) {
                        float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(a.out_pool) + qoff[u]);
                        *reinterpret_cast<float4*>(q) = make_float4(x[0], x[1], x[2], x[3]);
                        *reinterpret_cast<float4*>(q + 4) = make_float4(x[4], x[5], x[6], x[7]);
                    }
                }
                if constexpr (MODE == C32_SPLIT) {
                    half8 hi, lo;
                    split8(x, hi, lo);
                    *reinterpret_cast<half8*>(dst) = hi;
                    *reinterpret_cast<half8*>(dst + 16) = lo;
                } else {
                    bf16x8v b8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) b8[e] = (__bf16)x[e];
                    *reinterpret_cast<bf16x8v*>(dst) = b8;
                }
            }
        }
        if (j0 + G::NTHR * U >= nA) break;
    }
}

// ---- the MFMA loop of NTB tiles: taps x k-chunks of part PART, B operands through a ring of PD reads per tile -------------------
template <class G, int PART>
__device__ __forceinline__ void c32_mma(const uint8_t* smem, const int (&base)[G::NTB], const C32W<G>& w,
                                        f32x16 (&acc)[G::NTB][G::SPW]) {
    constexpr int NIT = G::TAPS * G::KCP, PD = NIT < 4 ? NIT : 4;
    uint4 ring[PD][G::NTB][G::NOP];
    auto rd = [&](const int it, const int slot) {
        const int tap = it / G::KCP, kcp = it % G::KCP;
        const int off = G::tap_off(tap) + G::kc_off(PART * G::KCP + kcp);
#pragma unroll
        for (int j = 0; j < G::NTB; ++j)
#pragma unroll
            for (int o = 0; o < G::NOP; ++o)
                ring[slot][j][o] = *reinterpret_cast<const uint4*>(smem + base[j] + off + 16 * o);
    };
#pragma unroll
    for (int it = 0; it < PD; ++it) rd(it, it);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = it % PD;
#pragma unroll
        for (int j = 0; j < G::NTB; ++j)
#pragma unroll
            for (int s = 0; s < G::SPW; ++s) {
                f32x16& c = acc[j][s];
                if constexpr (G::MODE == C32_SPLIT) {
                    const half8 Wh = __builtin_bit_cast(half8, w.v[s][it][0]), Wl = __builtin_bit_cast(half8, w.v[s][it][G::NOP - 1]);
                    const half8 Xh = __builtin_bit_cast(half8, ring[slot][j][0]), Xl = __builtin_bit_cast(half8, ring[slot][j][G::NOP - 1]);
                    c = mfma16(Wl, Xh, c);
                    c = mfma16(Wh, Xl, c);
                    c = mfma16(Wh, Xh, c);
                } else if constexpr (G::MODE == C32_BF16) {
                    c = mfma16_bf(__builtin_bit_cast(bf16x8v, w.v[s][it][0]), __builtin_bit_cast(bf16x8v, ring[slot][j][0]), c);
                } else {
                    c = mfma16(__builtin_bit_cast(half8, w.v[s][it][0]), __builtin_bit_cast(half8, ring[slot][j][0]), c);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        if (it + PD < NIT) rd(it + PD, slot);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- one layer for one member of a group: `a` is already the group's image range (a.nimg = its G images) ---------------------
// `w` holds the fragments of part 0 (c32_load_weights, issued by the caller before its barrier).
template <class G, bool RELU>
__device__ __forceinline__ void c32_run(const ConvArgs& a, uint8_t* smem, int member, C32W<G>& w) {
    using T = std::conditional_t<G::MODE == C32_NATIVE, half_t, float>;
    static_assert(G::KPARTS == 1, "k parts: not built yet");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int sg = wave % G::SG, tl = wave / G::SG;
    int sA, sB, nsb, rows;
    G::member_rows(member, a.nimg, sA, sB);
    G::sub_bands(sA, sB, nsb, rows);
    // accumulator init = bias of the lane's 16 output channels
    f32x16 bias[G::SPW];
#pragma unroll
    for (int s = 0; s < G::SPW; ++s) {
        const int cs = (sg * G::SPW + s) % G::CS;
        const float4* bp = reinterpret_cast<const float4*>(a.bias + 32 * cs + 16 * hi);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = bp[q];
            bias[s][4 * q] = b4.x; bias[s][4 * q + 1] = b4.y; bias[s][4 * q + 2] = b4.z; bias[s][4 * q + 3] = b4.w;
        }
    }
    const int lbase = G::lane_base(lane);
    C32_T(a, 0);
    for (int b = 0; b < nsb; ++b) {
        const int sb = sA + b * rows;
        const int R = (sB - sb) < rows ? (sB - sb) : rows;
        if (b > 0) __syncthreads();                        // everyone has finished reading the previous sub-band
        c32_stage<G>(a, smem, sb, R);
        C32_T(a, 1);
        __syncthreads();                                   // (the compiler waits for the LDS writes before the barrier)
        C32_T(a, 2);
        const int NT = G::n_tiles(R);
        for (int t0 = tl; t0 < NT; t0 += G::TL * G::NTB) {
            int tt[G::NTB], base[G::NTB];
            f32x16 acc[G::NTB][G::SPW];
#pragma unroll
            for (int j = 0; j < G::NTB; ++j) {
                tt[j] = t0 + G::TL * j;
                base[j] = (tt[j] < NT ? tt[j] : t0) * G::tile_step() + lbase;      // (a missing second tile repeats the first; not stored)
#pragma unroll
                for (int s = 0; s < G::SPW; ++s) acc[j][s] = bias[s];
            }
            c32_mma<G, 0>(smem, base, w, acc);
#pragma unroll
            for (int j = 0; j < G::NTB; ++j) {
                if (tt[j] >= NT) continue;
                const typename G::Out o = G::out_pixel(tt[j], n, sb, R);
                if (!o.valid) continue;
#pragma unroll
                for (int s = 0; s < G::SPW; ++s) {
                    const int slice = sg * G::SPW + s, sub = slice / G::CS, cs = slice % G::CS;
                    const size_t pix = (size_t)G::out_index(o.g, o.y, o.x, sub);
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = RELU ? relu(acc[j][s][r]) : acc[j][s][r];
                    T* dst = reinterpret_cast<T*>(a.out) + pix * G::COUT + 32 * cs + 16 * hi;
                    if constexpr (G::MODE == C32_NATIVE) {
                        half8 h0, h1;
#pragma unroll
                        for (int r = 0; r < 8; ++r) { h0[r] = (half_t)v[r]; h1[r] = (half_t)v[8 + r]; }
                        *reinterpret_cast<half8*>(dst) = h0;
                        *reinterpret_cast<half8*>(dst + 8) = h1;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    }
                    if constexpr (G::KIND == CONV1) {
                        if (a.out_nchw) {
                            float* q = a.out_nchw + ((size_t)o.g * G::COUT + 32 * cs + 16 * hi) * (G::H * G::W) + o.y * G::W + o.x;
#pragma unroll
                            for (int r = 0; r < 16; ++r) q[(size_t)r * (G::H * G::W)] = v[r];
                        }
                    }
                }
            }
        }
        C32_T(a, 3);
    }
}

// the image range [img0, img0 + n) of one layer's operands
template <class G>
__device__ __forceinline__ ConvArgs c32_image_range(ConvArgs a, int img0, int n) {
    if (img0 != 0) a.trace_id = -2;                  // (diagnostic builds trace the group that holds image 0)
    const size_t es = G::ES;
    a.in0 = reinterpret_cast<const char*>(a.in0) + (size_t)img0 * G::IH * G::IW * G::C0 * es;
    if (a.in1) a.in1 = reinterpret_cast<const char*>(a.in1) + (size_t)img0 * G::IH * G::IW * G::C1 * es;
    a.out = reinterpret_cast<char*>(a.out) + (size_t)img0 * G::OH * G::OW * G::COUT * es;
    if (a.out_pool) a.out_pool = reinterpret_cast<char*>(a.out_pool) + (size_t)img0 * G::H * G::W * G::C0 * es;
    if (a.out_nchw) a.out_nchw += (size_t)img0 * G::COUT * G::H * G::W;
    a.nimg = n;
    return a;
}

// group q of nq: a contiguous, balanced share of the images (the persistent kernel's rule)
__device__ __forceinline__ void c32_group_images(int q, int nq, int nimg, int& img0, int& per) {
    img0 = (q * nimg + nq - 1) / nq;
    per = ((q + 1) * nimg + nq - 1) / nq - img0;
}

// one layer as its own launch: nq groups x 8 workgroups (stage probes, A/B runs; the product path is the persistent kernel)
template <class G, bool RELU>
__global__ __launch_bounds__(C32_NW * 64) void conv32_kernel(ConvArgs a, int nq) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int q = (int)blockIdx.x / C32_GROUP, member = (int)blockIdx.x % C32_GROUP;
    int img0, per;
    c32_group_images(q, nq, a.nimg, img0, per);
    if (per == 0) return;
    const ConvArgs ar = c32_image_range<G>(a, img0, per);
    C32W<G> w;
    c32_load_weights<G>(ar, w, 0);
    c32_run<G, RELU>(ar, smem, member, w);
}

inline int c32_groups(int nimg) {                     // groups of a launch: 8 per slot, up to 4 slots per XCD (as unet_mega_kernel)
    const int slots = (nimg + 7) / 8 < 4 ? (nimg + 7) / 8 : 4;
    return 8 * slots;
}

template <class G, bool RELU>
inline int launch_conv32(const ConvArgs& a, hipStream_t s) {
    const int nq = c32_groups(a.nimg);
    auto kern = conv32_kernel<G, RELU>;
    giga::dyn_lds_once(reinterpret_cast<const void*>(kern), C32_LDS);
    GIGA_LAUNCH(kern, dim3(nq * C32_GROUP), dim3(C32_NW * 64), C32_LDS, s, a, nq);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
