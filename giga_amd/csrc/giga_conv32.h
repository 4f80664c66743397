// conv32: the LDS-image-resident 32x32x16 convolution of the f16-class U-Net (f16, f16x3 split and bf16 arithmetic).
// Geometry, the reasons for it and the reference citations: giga_conv32_geom.h.  This file is the gfx950 side:
//   c32_fill   : LDS-DMA of the member's weight fragments into LDS bytes [0, WBYTES) -- one copy per workgroup; issued before the
//                group barrier of the persistent kernel, so it lands while the barrier is waited for;
//   c32_stage  : the workgroup copies its haloed sub-band into LDS behind the weights (16-byte vectors, every load of a thread in
//                flight before its first LDS write, pad columns / rows written as zeros; 2x2 max-pool, f16x3 split or bf16
//                rounding applied on the way);
//   c32_mma    : a wave's register tile of NTB tiles x SPW slices: per (tap, k-chunk) SPW + NTB ds_read_b128 at `base +
//                immediate` feed NTB * SPW MFMAs; a ring of three steps is in flight;
//   epilogue   : a lane holds 16 consecutive output channels of one pixel: bias (accumulator init), ReLU, rounding, one or
//                two 16-byte stores per 8 channels.
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

// the member's WFR fragments of channel part `part` -> LDS bytes [0, WBYTES): 1 KiB per wave instruction (all waves take part)
// The layer's biases ride along (part 0; float `boff` of the bias block behind the images): a register tile starts from them, and
// fetching them from memory at the start of every tile costs a global-load latency per tile (1.5-2 k clocks: as long as the
// MFMAs of a one-scene tile).
__device__ __forceinline__ float* c32_bias_lds() {
    extern __shared__ __attribute__((aligned(16))) uint8_t c32_dyn_lds[];                  // (every dynamic LDS array starts at the same byte)
    return reinterpret_cast<float*>(c32_dyn_lds + C32_LDS);
}
template <class G>
__device__ __forceinline__ void c32_fill(const ConvArgs& a, uint8_t* smem, int member, int part = 0, int boff = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sgm = G::member_sgm(member);
    const uint8_t* wsrc = a.w + lane * 16;
    for (int c = wave; c < G::WFR; c += C32_NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)G::fill_src(c, sgm, part) * FRAG),
                                         (__attribute__((address_space(3))) void*)(smem + c * FRAG), 16, 0, 0);
    static_assert(G::COUT * 4 <= C32_BIAS_BYTES, "bias block");
    if (part == 0 && wave == C32_NW - 1) {
#pragma unroll
        for (int c = 0; c < (G::COUT + 63) / 64; ++c)
            if (64 * c + lane < G::COUT)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias + 64 * c + lane),
                                                 (__attribute__((address_space(3))) void*)(c32_bias_lds() + boff + 64 * c), 4, 0, 0);
    }
}

// ---- staging: the haloed sub-band [sb - HALO, sb + R + HALO) x P pixels -> LDS ------------------------------------------------
// Thread t takes items t, t + 256, ... of the sub-band's real rows (giga_conv32_geom.h: Cur -- source and LDS offsets advance by
// adds).  A chunk issues ALL its loads (8 / 16 / 24 items of 16 bytes per thread, chosen by what is left) before the first LDS
// write; the zero pixels are written under the latency of the first chunk.
template <class G>
struct C32Stage {
    static constexpr int MODE = G::MODE, ES = G::ES;
    static constexpr int NPOS = G::POOLIN ? 4 : 1;                  // source pixels per staged pixel
    static constexpr int VPI = MODE == C32_NATIVE ? 1 : 2;          // 16-byte source vectors per item (8 channels)
    static constexpr int UMAX = 24 / (VPI * NPOS);                  // items per thread in flight at most
    const ConvArgs& a;
    uint8_t* smem;
    const char* src;                                                // this thread's source tensor (concat: in0 or in1)
    uint32_t pixb, chb, qchb;                                       // bytes per source pixel, byte offset of the thread's 8 channels (source / pooled copy)
    int sb, R, nA, j;
    typename G::Cur k;
    bool zero_pending;

    __device__ __forceinline__ C32Stage(const ConvArgs& a_, uint8_t* smem_, int sb_, int R_, int part) : a(a_), smem(smem_), sb(sb_), R(R_) {
        const int tid = threadIdx.x;
        int rrA, rrB;
        G::real_rows(sb, R, a.nimg, rrA, rrB);
        nA = (rrB - rrA) * G::RI;
        const int ch0 = G::thr_ch(tid);
        // KP == 1: a concatenated pixel is [C0 channels of in0 | C1 channels of in1]; KP > 1: the part names its tensor / channel range
        const bool first = G::KP > 1 ? G::part_tensor(part) == 0 : (G::C1 == 0 || ch0 < G::C0);
        src = reinterpret_cast<const char*>(first ? a.in0 : a.in1);
        pixb = (uint32_t)(first ? G::C0 : G::C1) * ES;
        chb = (uint32_t)(G::KP > 1 ? G::part_ch0(part) + ch0 : (first ? ch0 : ch0 - G::C0)) * ES;
        qchb = (uint32_t)ch0 * ES;
        k = G::cur_init(rrA, tid, sb);
        j = tid;
        zero_pending = G::HALO != 0;
    }
    __device__ __forceinline__ void zeros() {
        zero_pending = false;
        const int nq = G::n_buf_pixels(R);
        for (int q = threadIdx.x; q < nq; q += G::NTHR)
            if (G::pad_pixel(q, sb, a.nimg)) {
#pragma unroll
                for (int e = 0; e < G::IPP * G::ILB / 16; ++e)
                    *reinterpret_cast<uint4*>(smem + G::WBYTES + q * G::PS + 16 * e) = make_uint4(0, 0, 0, 0);
            }
    }
    template <int U>
    __device__ __forceinline__ void chunk() {
        uint4 v[U][NPOS][VPI];
        int lds[U];
        uint32_t qoff[U];
        bool ok[U], own[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = j < nA;
            lds[u] = k.lds;
#pragma unroll
            for (int q = 0; q < NPOS; ++q) {
                const uint32_t off = ok[u] ? __umul24((uint32_t)G::cur_src_pixel(k, q), pixb) + chb : 0u;   // (clamped: every load is issued)
#pragma unroll
                for (int e = 0; e < VPI; ++e) v[u][q][e] = *reinterpret_cast<const uint4*>(src + off + 16 * e);
            }
            if constexpr (G::POOLIN) {
                own[u] = ok[u] && G::cur_own(k, sb, R);
                qoff[u] = __umul24((uint32_t)k.spix, (uint32_t)G::C0 * ES) + qchb;
            }
            G::cur_next(k);
            j += G::NTHR;
        }
        if (zero_pending) zeros();                           // under the latency of the loads
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
    }
    __device__ __forceinline__ void run() {
        constexpr int U1 = UMAX >= 3 ? UMAX / 3 : 1, U2 = UMAX >= 3 ? 2 * UMAX / 3 : (UMAX >= 2 ? 2 : 1);
        int rem = (nA + G::NTHR - 1) / G::NTHR;              // steps of the slowest thread (uniform)
        while (rem > 0) {
            if (rem > U2 && UMAX > U2) { chunk<UMAX>(); rem -= UMAX; }
            else if (rem > U1 && U2 > U1) { chunk<U2>(); rem -= U2; }
            else { chunk<U1>(); rem -= U1; }
        }
        if (zero_pending) zeros();
    }
};
template <class G>
__device__ __forceinline__ void c32_stage(const ConvArgs& a, uint8_t* smem, int sb, int R, int part = 0) {
    C32Stage<G> st(a, smem, sb, R, part);
    st.run();
}

// ---- the MFMA loop of a register tile: NTB tiles x SPW slices, taps x k-chunks; operands through a ring of PD steps ------------
// wl: this lane's LDS byte offset of the first A fragment of the slice pass (lane * 16 + pass base); base[j]: its B offset of tile j
template <class G, int NTB>
__device__ __forceinline__ void c32_mma(const uint8_t* smem, const int wl, const int (&base)[NTB], f32x16 (&acc)[NTB][G::SPW]) {
    // ring depth: operands are requested ~400 clocks of MFMA work ahead (an LDS read under load takes 150-300 clocks; a step
    // of a small register tile is a single 32-clock MFMA, and with one wave per SIMD at small batches nobody else hides it)
    constexpr int MPS = NTB * G::SPW * (G::MODE == C32_SPLIT ? 3 : 1);
    constexpr int PDW = (12 + MPS - 1) / MPS < 3 ? 3 : (12 + MPS - 1) / MPS > 8 ? 8 : (12 + MPS - 1) / MPS;
    constexpr int NIT = G::NIT, PD = NIT < PDW ? NIT : PDW;
    uint4 ra[PD][G::SPW][G::NOP], rb[PD][NTB][G::NOP];
    auto rd = [&](const int it, const int slot) {
        const int off = G::tap_off(it / G::KCP) + G::kc_off(it % G::KCP);
#pragma unroll
        for (int s = 0; s < G::SPW; ++s)
#pragma unroll
            for (int o = 0; o < G::NOP; ++o) ra[slot][s][o] = *reinterpret_cast<const uint4*>(smem + wl + G::w_off(s, it, o));
#pragma unroll
        for (int j = 0; j < NTB; ++j)
#pragma unroll
            for (int o = 0; o < G::NOP; ++o) rb[slot][j][o] = *reinterpret_cast<const uint4*>(smem + base[j] + off + 16 * o);
    };
#pragma unroll
    for (int it = 0; it < PD; ++it) rd(it, it);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = it % PD;
#pragma unroll
        for (int j = 0; j < NTB; ++j)
#pragma unroll
            for (int s = 0; s < G::SPW; ++s) {
                f32x16& c = acc[j][s];
                if constexpr (G::MODE == C32_SPLIT) {
                    const half8 Wh = __builtin_bit_cast(half8, ra[slot][s][0]), Wl = __builtin_bit_cast(half8, ra[slot][s][G::NOP - 1]);
                    const half8 Xh = __builtin_bit_cast(half8, rb[slot][j][0]), Xl = __builtin_bit_cast(half8, rb[slot][j][G::NOP - 1]);
                    c = mfma16(Wl, Xh, c);
                    c = mfma16(Wh, Xl, c);
                    c = mfma16(Wh, Xh, c);
                } else if constexpr (G::MODE == C32_BF16) {
                    c = mfma16_bf(__builtin_bit_cast(bf16x8v, ra[slot][s][0]), __builtin_bit_cast(bf16x8v, rb[slot][j][0]), c);
                } else {
                    c = mfma16(__builtin_bit_cast(half8, ra[slot][s][0]), __builtin_bit_cast(half8, rb[slot][j][0]), c);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        if (it + PD < NIT) rd(it + PD, slot);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// a register tile: NTB tiles (t0 ..) x the SPW slices of slice pass sp
template <class G, bool RELU, int NTB>
struct C32Tile {
    using T = std::conditional_t<G::MODE == C32_NATIVE, half_t, float>;
    f32x16 acc[NTB][G::SPW];
    int base[NTB];
    // ioff: byte offset of the staged image behind the weights (a fused pair keeps two images)
    __device__ __forceinline__ void init(const ConvArgs& a, int sgm, int sp, int t0, int NT, int ioff = 0, int boff = 0) {
        const int lane = threadIdx.x & 63, hi = lane >> 5;
        const int lbase = G::lane_base(lane) + ioff;
#pragma unroll
        for (int j = 0; j < NTB; ++j) base[j] = (t0 + j < NT ? t0 + j : t0) * G::tile_step() + lbase;   // (a missing tile repeats the first; not stored)
#pragma unroll
        for (int s = 0; s < G::SPW; ++s) {                 // accumulator init = bias of the lane's 16 output channels
            const int cs = (sgm * G::SPM + sp * G::SPW + s) % G::CS;
            const float4* bp = reinterpret_cast<const float4*>(c32_bias_lds() + boff + 32 * cs + 16 * hi);   // (LDS: c32_fill)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = bp[q];
#pragma unroll
                for (int j = 0; j < NTB; ++j) {
                    acc[j][s][4 * q] = b4.x; acc[j][s][4 * q + 1] = b4.y; acc[j][s][4 * q + 2] = b4.z; acc[j][s][4 * q + 3] = b4.w;
                }
            }
        }
    }
    __device__ __forceinline__ void mma(const uint8_t* smem, int sp) {
        c32_mma<G, NTB>(smem, (int)(threadIdx.x & 63) * 16 + G::w_off(sp * G::SPW, 0, 0), base, acc);
    }
    __device__ __forceinline__ void store(const ConvArgs& a, int sgm, int sp, int t0, int NT, int sb, int R) {
        const int lane = threadIdx.x & 63, n = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            if (t0 + j >= NT) continue;
            const typename G::Out o = G::out_pixel(t0 + j, n, sb, R);
            if (!o.valid) continue;
#pragma unroll
            for (int s = 0; s < G::SPW; ++s) {
                const int slice = sgm * G::SPM + sp * G::SPW + s, sub = slice / G::CS, cs = slice % G::CS;
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
};

// the tiles of one staged sub-band (KP == 1): batches of NTB tiles dealt over the waves, all slice passes of the member per batch
template <class G, bool RELU, int NTB>
__device__ __forceinline__ void c32_tiles(const ConvArgs& a, const uint8_t* smem, int sgm, int NT, int sb, int R, int ioff = 0, int boff = 0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbatch = (NT + NTB - 1) / NTB;
    // a task = (batch, slice pass); tasks are dealt over the waves, so that the passes of a small sub-band (the ConvTranspose
    // layers of a single scene: one batch, four passes) run side by side instead of one after the other on one wave
    for (int task = wave; task < nbatch * G::NSP; task += C32_NW) {
        const int bi = task / G::NSP, sp = task - bi * G::NSP;
        C32Tile<G, RELU, NTB> t;
        t.init(a, sgm, sp, bi * NTB, NT, ioff, boff);
        t.mma(smem, sp);
        t.store(a, sgm, sp, bi * NTB, NT, sb, R);
    }
}
// a sub-band whose channels are walked in KP parts: ONE register tile per wave (RBMAX keeps it to that), alive across the parts;
// each part has its own weight fill and its own staged image
template <class G, bool RELU, int NTB>
__device__ __forceinline__ void c32_parts(const ConvArgs& a, uint8_t* smem, int member, bool filled, int NT, int sb, int R) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sgm = G::member_sgm(member);
    const bool mine = wave * NTB < NT;
    C32Tile<G, RELU, NTB> t;
#pragma unroll 1
    for (int part = 0; part < G::KP; ++part) {
        if (part > 0 || !filled) {
            __syncthreads();                               // everyone has finished with the previous weights and image
            c32_fill<G>(a, smem, member, part);
        }
        c32_stage<G>(a, smem, sb, R, part);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (part == 0) t.init(a, sgm, 0, wave * NTB, NT);  // (the biases arrived with part 0's weights)
        if (mine) t.mma(smem, 0);
    }
    if (mine) t.store(a, sgm, 0, wave * NTB, NT, sb, R);
}

// ---- one layer for one member of a group: `a` is already the group's image range (a.nimg = its G images); the member's
// weights (part 0) are on their way into LDS (c32_fill, issued by the caller before its barrier).
template <class G, bool RELU>
__device__ __forceinline__ void c32_tiles_any(const ConvArgs& a, const uint8_t* smem, int sgm, int NT, int sb, int R, int ioff, int boff = 0) {
    const int ntb = G::batch_tiles(NT);
    if constexpr (G::SPW > 1 && G::KP == 1) {              // few tasks: one-slice register tiles, twice as many tasks (bit-identical)
        if ((NT + ntb - 1) / ntb * G::NSP <= C32_NW / 2) {
            c32_tiles_any<typename G::Thin, RELU>(a, smem, sgm, NT, sb, R, ioff, boff);
            return;
        }
    }
    if (ntb == 1) c32_tiles<G, RELU, 1>(a, smem, sgm, NT, sb, R, ioff, boff);
    else if (ntb == 2) c32_tiles<G, RELU, 2>(a, smem, sgm, NT, sb, R, ioff, boff);
    else if constexpr (G::NTBM >= 3) c32_tiles<G, RELU, 3>(a, smem, sgm, NT, sb, R, ioff, boff);
}
// ---- fused pair A -> B (giga_conv32_geom.h: C32Pair): A's epilogue writes B's input image in LDS ------------------------------------
// the register tile of A (sgm = 0): values -> B's image (every row of A's sub-band) and, for the rows the member owns, memory
template <class GA, class GB, bool RELU, int NTB>
__device__ __forceinline__ void c32_store_mid(const C32Tile<GA, RELU, NTB>& t, const ConvArgs& a, uint8_t* mid, int sp, int t0, int NT,
                                              int sbA, int RA, int R) {
    using T = std::conditional_t<GA::MODE == C32_NATIVE, half_t, float>;
    using PR = C32Pair<GA, GB>;
    const int lane = threadIdx.x & 63, n = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        if (t0 + j >= NT) continue;
        const typename GA::Out o = GA::out_pixel(t0 + j, n, sbA, RA);
        if (!o.valid) continue;
        const bool own = o.orow >= 1 && o.orow <= R;           // (A's sub-band = the member's rows + one above and one below)
#pragma unroll
        for (int s = 0; s < GA::SPW; ++s) {
            const int cs = (sp * GA::SPW + s) % GA::CS;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = RELU ? relu(t.acc[j][s][r]) : t.acc[j][s][r];
            uint8_t* d = mid + PR::mid_off(o.orow, o.x, 4 * cs + 2 * hi);
            T* dst = reinterpret_cast<T*>(a.out) + (size_t)GA::out_index(o.g, o.y, o.x, 0) * GA::COUT + 32 * cs + 16 * hi;
            if constexpr (GA::MODE == C32_NATIVE) {
                half8 h0, h1;
#pragma unroll
                for (int r = 0; r < 8; ++r) { h0[r] = (half_t)v[r]; h1[r] = (half_t)v[8 + r]; }
                *reinterpret_cast<half8*>(d) = h0;
                *reinterpret_cast<half8*>(d + 16) = h1;
                if (own) { *reinterpret_cast<half8*>(dst) = h0; *reinterpret_cast<half8*>(dst + 8) = h1; }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float x8[8] = {v[8 * k], v[8 * k + 1], v[8 * k + 2], v[8 * k + 3], v[8 * k + 4], v[8 * k + 5], v[8 * k + 6], v[8 * k + 7]};
                    if constexpr (GA::MODE == C32_SPLIT) {
                        half8 xh, xl;
                        split8(x8, xh, xl);
                        *reinterpret_cast<half8*>(d + 32 * k) = xh;
                        *reinterpret_cast<half8*>(d + 32 * k + 16) = xl;
                    } else {
                        bf16x8v b8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) b8[e] = (__bf16)x8[e];
                        *reinterpret_cast<bf16x8v*>(d + 16 * k) = b8;
                    }
                }
                if (own) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
            }
        }
    }
}
template <class GA, class GB, bool RELU, int NTB>
__device__ __forceinline__ void c32_tiles_mid(const ConvArgs& a, uint8_t* smem, int ioff, uint8_t* mid, int NT, int sbA, int RA, int R) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbatch = (NT + NTB - 1) / NTB;
    for (int task = wave; task < nbatch * GA::NSP; task += C32_NW) {       // (batch, slice pass) tasks over the waves, as c32_tiles
        const int bi = task / GA::NSP, sp = task - bi * GA::NSP;
        C32Tile<GA, RELU, NTB> t;
        t.init(a, 0, sp, bi * NTB, NT, ioff);
        t.mma(smem, sp);
        c32_store_mid<GA, GB, RELU, NTB>(t, a, mid, sp, bi * NTB, NT, sbA, RA, R);
    }
}
template <class GA, class GB, bool RELU>
__device__ __forceinline__ void c32_tiles_mid_any(const ConvArgs& a, uint8_t* smem, int ioff, uint8_t* mid, int NT, int sbA, int RA, int R) {
    const int ntb = GA::batch_tiles(NT);
    if constexpr (GA::SPW > 1) {                           // few tasks: one-slice register tiles (see c32_tiles_any)
        if ((NT + ntb - 1) / ntb * GA::NSP <= C32_NW / 2) {
            c32_tiles_mid_any<typename GA::Thin, GB, RELU>(a, smem, ioff, mid, NT, sbA, RA, R);
            return;
        }
    }
    if (ntb == 1) c32_tiles_mid<GA, GB, RELU, 1>(a, smem, ioff, mid, NT, sbA, RA, R);
    else if (ntb == 2) c32_tiles_mid<GA, GB, RELU, 2>(a, smem, ioff, mid, NT, sbA, RA, R);
    else if constexpr (GA::NTBM >= 3) c32_tiles_mid<GA, GB, RELU, 3>(a, smem, ioff, mid, NT, sbA, RA, R);
}
// can A -> B run as a fused pair?  (same-resolution 3x3 layers, whole weights of both in LDS, room for sub-bands of 6 rows and more)
template <class GA, class GB>
constexpr bool c32_pair_ok() {
    if constexpr (GA::KIND == CONV3 && GB::KIND == CONV3 && GA::H == GB::H && GA::COUT == GB::CIN && GB::C1 == 0 && !GB::POOLIN &&
                  GA::SGM == 1 && GB::SGM == 1 && GA::KP == 1 && GB::KP == 1)
        return (C32_LDS - GA::WBYTES - GB::WBYTES - (4 * GA::ROWB + 2 * GB::ROWB + C32_TAIL * (GA::PS + GB::PS))) / (GA::ROWB + GB::ROWB) >= 6;
    else
        return false;
}
// both weight sets: A's at LDS byte 0, B's behind them
template <class GA, class GB>
__device__ __forceinline__ void c32_fill_pair(const ConvArgs& a, const ConvArgs& b, uint8_t* smem, int member) {
    c32_fill<GA>(a, smem, member, 0, 0);
    c32_fill<GB>(b, smem + GA::WBYTES, member, 0, 64);
}
// layers A and B for one member of a group, no group barrier between them
template <class GA, class GB, bool RELU_A, bool RELU_B>
__device__ __forceinline__ void c32_run_pair(const ConvArgs& a, const ConvArgs& b, uint8_t* smem, int member) {
    using PR = C32Pair<GA, GB>;
    int sA, sB, nsb, rows;
    GB::member_rows(member, b.nimg, sA, sB);
    PR::sub_bands(sA, sB, nsb, rows);
    C32_T(a, 0);
    for (int bi = 0; bi < nsb; ++bi) {
        const int sb = sA + bi * rows;
        const int R = (sB - sb) < rows ? (sB - sb) : rows;
        uint8_t* mid = smem + PR::mid0(R);
        if (bi > 0) __syncthreads();                       // everyone has finished reading the previous sub-band's images
        {
            C32Stage<GA> st(a, smem + PR::WB, sb - 1, R + 2, 0);      // A's input: the member's rows with a halo of two
            st.run();
            C32Stage<GB> zb(b, mid - GB::WBYTES, sb, R, 0);           // the zero rows / columns of B's image
            zb.zeros();
        }
        C32_T(a, 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's share of both weight fills has landed
        __syncthreads();
        C32_T(a, 2);
        c32_tiles_mid_any<GA, GB, RELU_A>(a, smem, PR::WB, mid, GA::n_tiles(R + 2), sb - 1, R + 2, R);
        C32_T(a, 3);
        __syncthreads();                                   // B's image is complete
        C32_T(b, 2);
        c32_tiles_any<GB, RELU_B>(b, smem + PR::WA, 0, GB::n_tiles(R), sb, R, PR::imgA_bytes(R), 64);
        C32_T(b, 3);
    }
}

template <class G, bool RELU>
__device__ __forceinline__ void c32_run(const ConvArgs& a, uint8_t* smem, int member) {
    const int sgm = G::member_sgm(member);
    int sA, sB, nsb, rows;
    G::member_rows(member, a.nimg, sA, sB);
    C32_T(a, 0);
    if constexpr (G::KP > 1) {
        G::sub_bands(sA, sB, nsb, rows);
        for (int b = 0; b < nsb; ++b) {
            const int sb = sA + b * rows;
            const int R = (sB - sb) < rows ? (sB - sb) : rows;
            const int NT = G::n_tiles(R);
            const int ntb = G::batch_tiles(NT);
            if (ntb == 1) c32_parts<G, RELU, 1>(a, smem, member, b == 0, NT, sb, R);
            else if (ntb == 2) c32_parts<G, RELU, 2>(a, smem, member, b == 0, NT, sb, R);
            else if constexpr (G::NTBM >= 3) c32_parts<G, RELU, 3>(a, smem, member, b == 0, NT, sb, R);
        }
    } else {
        G::sub_bands(sA, sB, nsb, rows);
        for (int b = 0; b < nsb; ++b) {
            const int sb = sA + b * rows;
            const int R = (sB - sb) < rows ? (sB - sb) : rows;
            if (b > 0) __syncthreads();                    // everyone has finished reading the previous sub-band
            c32_stage<G>(a, smem, sb, R);
            C32_T(a, 1);
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's share of the weight fill has landed
            __syncthreads();                               // (the compiler waits for the LDS writes before the barrier)
            C32_T(a, 2);
            c32_tiles_any<G, RELU>(a, smem, sgm, G::n_tiles(R), sb, R, 0);
            C32_T(a, 3);
        }
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
    c32_fill<G>(ar, smem, member);
    c32_run<G, RELU>(ar, smem, member);
}

inline int c32_groups(int nimg) {                     // groups of a launch: 8 per slot, up to 4 slots per XCD (as unet_mega_kernel)
    const int slots = (nimg + 7) / 8 < 4 ? (nimg + 7) / 8 : 4;
    return 8 * slots;
}

template <class G, bool RELU>
inline int launch_conv32(const ConvArgs& a, hipStream_t s) {
    const int nq = c32_groups(a.nimg);
    auto kern = conv32_kernel<G, RELU>;
    giga::dyn_lds_once(reinterpret_cast<const void*>(kern), C32_LDS_TOTAL);
    GIGA_LAUNCH(kern, dim3(nq * C32_GROUP), dim3(C32_NW * 64), C32_LDS_TOTAL, s, a, nq);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
