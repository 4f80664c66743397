// bf16 images of the decoder heads for the bf16 TRAINING step (BASELINE c5; csrc/giga_decoder_train16.hip): layout and the
// element-by-element derivation from the fp32 images of the same blobs, shared by the host packer (giga_pack.cpp) and the
// device-side derivation that follows every repack (giga_derive_bf16_fragments).  Host and device evaluate the SAME function on the
// same fp32 words, so the two images are equal byte for byte.
//
// Forward image (in the forward blob, PackOff::dect[h]): the fragment order and k-slot maps of the f16 image (giga_layout.h
// DEC16_*; v_mfma_f32_32x32x16_bf16 has the operand layout of the f16 instruction) with bf16 elements, followed by the fp32 C table:
//   block b: fragments 11b .. 11b+5 fc_c chunks, 11b+6 aux (fc_p as hi/lo pairs in block 0, the folded biases as hi/lo pairs),
//            11b+7,8 fc_0, 11b+9,10 fc_1;  55 aux (bias of the last fc_1), 56,57 fc_out;  chunk 58: 5 x 32 fc_0 biases + fc_out bias.
// Backward image (in the backward blob, BwdPackOff::dect[h]): A operands of the TRANSPOSED matrices for the gradient chain,
//   block b: 10b + 2 rb + c   Wc_b^T, rows = input features 32 rb .. 32 rb + 31, k-chunk c;  10b+6,7 W0_b^T;  10b+8,9 W1_b^T
//   chunk 50: fc_out.weight as fp32 [4][32], rounded to bf16 (the value the forward multiplied with).
// A-operand fragment of a transposed 32x32 matrix M^T, chunk c: lane (i, hi), slot j -> M[drow(8c+j, hi)][i]: the contraction
// index runs over the D-register order of the gradient that feeds it, as in the forward chain.
#pragma once
#include <hip/hip_runtime.h>

#include "giga_layout.h"

namespace giga {

constexpr int DECT_FWD_FRAGS = DEC16_FRAGS;                            // 58
constexpr size_t DECT_FWD_BYTES = DEC16_BYTES;                         // 59 KiB
constexpr int DECT_BWD_FRAGS = NBLK * 10;                              // 50
constexpr size_t DECT_BWD_BYTES = (size_t)(DECT_BWD_FRAGS + 1) * FRAG; // 51 KiB

// fp32 -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32 for finite values), and back
__host__ __device__ inline uint16_t dect_bf(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float dect_f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// element (fragment idx, lane, slot j) of the forward image from the fp32 forward image `f` of the head (DEC32 layout, giga_pack.cpp
// pack_head32: per block 12 fc_c fragments, 1 aux, 4 fc_0, 4 fc_1; tail 1 aux, 4 fc_out)
__host__ __device__ inline uint16_t dect_fwd_elem(const float* f, int idx, int lane, int j) {
    const int n = lane & 31, hi = lane >> 5;
    const int b = idx < 55 ? idx / 11 : NBLK, t = idx < 55 ? idx % 11 : idx - 55 + 6;      // tail: t = 6 aux, 7 / 8 fc_out
    const int base = 21 * b;                                   // first fp32 fragment of the block (tail: 105 = aux, 106.. fc_out)
    auto F = [&](int frag, int ln, int jj) { return f[((size_t)frag * 64 + ln) * 4 + jj]; };
    if (t < 6) {                                               // fc_c chunk t: k-slot (hi, j) = feature (t/2)*32 + (t%2)*16 + 8hi + j
        const int tt = (t % 2) * 16 + 8 * hi + j, m = 16 * (t / 2) + (tt & 15);
        return dect_bf(F(base + m / 4, (tt >> 4) * 32 + n, m % 4));
    }
    if (t == 6) {                                              // aux: [Wp x p_hi (3), bias_hi, Wp x p_lo (3), bias_lo] | hi = 1: [Wp_lo x p_hi (3), 0..]
        const int af = b < NBLK ? base + 12 : 105;
        if (j < 3 || (hi == 0 && j >= 4 && j < 7)) {
            const int k = j < 3 ? j : j - 4;
            const float w = k == 0 ? F(af, n, 0) : k == 1 ? F(af, 32 + n, 0) : F(af, n, 1);
            const uint16_t wh = dect_bf(w);
            return hi == 0 ? wh : dect_bf(w - dect_f(wh));
        }
        if (hi == 1) return 0;
        const float bias = F(af, 32 + n, 1) + F(af, n, 2);
        const uint16_t bh = dect_bf(bias);
        return j == 3 ? bh : dect_bf(bias - dect_f(bh));
    }
    // dense 32 x 32 (fc_0: t = 7, 8; fc_1: t = 9, 10; tail fc_out: t = 7, 8): chunk c slot j = fp32 fragment 2c + j/4, slot j%4
    const int c = b < NBLK ? (t - 7) & 1 : t - 7;
    const int d0 = b < NBLK ? (t < 9 ? base + 13 : base + 17) : 106;
    return dect_bf(F(d0 + 2 * c + j / 4, lane, j % 4));
}

// element of the backward image from the fp32 backward image `g` of the head (DECB layout, giga_pack.cpp pack_head_bwd: per block
// Wc^T 3 x 4 fragments, W0^T 4, W1^T 4)
__host__ __device__ inline uint16_t dect_bwd_elem(const float* g, int idx, int lane, int j) {
    const int b = idx / 10, t = idx % 10;
    const int f0 = 20 * b + (t < 6 ? 4 * (t / 2) : t < 8 ? 12 : 16), c = t & 1;
    return dect_bf(g[((size_t)(f0 + 2 * c + j / 4) * 64 + lane) * 4 + j % 4]);
}

// host: fill both images of one head from its fp32 images
inline void dect_pack_fwd_host(const uint8_t* dec32_image, uint8_t* dst) {
    const float* f = reinterpret_cast<const float*>(dec32_image);
    uint16_t* o = reinterpret_cast<uint16_t*>(dst);
    for (int idx = 0; idx < DECT_FWD_FRAGS; ++idx)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) o[((size_t)idx * 64 + lane) * 8 + j] = dect_fwd_elem(f, idx, lane, j);
    const float* ctab = f + (size_t)DEC32_FRAGS * 256;
    float* oc = reinterpret_cast<float*>(dst + (size_t)DECT_FWD_FRAGS * FRAG);
    for (int i = 0; i < (NBLK + 1) * CD; ++i) oc[i] = ctab[i];
}
inline void dect_pack_bwd_host(const uint8_t* decb_image, uint8_t* dst) {
    const float* g = reinterpret_cast<const float*>(decb_image);
    uint16_t* o = reinterpret_cast<uint16_t*>(dst);
    for (int idx = 0; idx < DECT_BWD_FRAGS; ++idx)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) o[((size_t)idx * 64 + lane) * 8 + j] = dect_bwd_elem(g, idx, lane, j);
    const float* wout = g + (size_t)DECB_FRAGS * 256;
    float* ow = reinterpret_cast<float*>(dst + (size_t)DECT_BWD_FRAGS * FRAG);
    for (int i = 0; i < 4 * CD; ++i) ow[i] = dect_f(dect_bf(wout[i]));
}

// device: one 64-lane block of giga_capi.hip's derive_all_kernel per 1-KiB chunk of a head's two images:
// chunk < 58 forward fragment, 58 C table, 59 .. 108 backward fragment, 109 Wout
__device__ inline void dect_derive_block(uint8_t* fwd_blob, uint8_t* bwd_blob, int chunk, int h, size_t dec32_0, size_t dec32_stride,
                                  size_t dectf_0, size_t dectf_stride, size_t decb_0, size_t decb_stride, size_t dectb_0,
                                  size_t dectb_stride) {
    const int lane = threadIdx.x;
    if (chunk < 59) {
        if (!fwd_blob) return;
        const float* f = reinterpret_cast<const float*>(fwd_blob + dec32_0 + h * dec32_stride);
        uint8_t* dst = fwd_blob + dectf_0 + h * dectf_stride + (size_t)chunk * FRAG;
        if (chunk < DECT_FWD_FRAGS) {
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dect_fwd_elem(f, chunk, lane, j);
            *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(v[0] | (unsigned)v[1] << 16, v[2] | (unsigned)v[3] << 16,
                                                                    v[4] | (unsigned)v[5] << 16, v[6] | (unsigned)v[7] << 16);
        } else {
            const float* ctab = f + (size_t)DEC32_FRAGS * 256;
            float* oc = reinterpret_cast<float*>(dst);
            for (int i = lane; i < (NBLK + 1) * CD; i += 64) oc[i] = ctab[i];
        }
    } else {
        if (!bwd_blob) return;
        const int c2 = chunk - 59;
        const float* g = reinterpret_cast<const float*>(bwd_blob + decb_0 + h * decb_stride);
        uint8_t* dst = bwd_blob + dectb_0 + h * dectb_stride + (size_t)c2 * FRAG;
        if (c2 < DECT_BWD_FRAGS) {
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dect_bwd_elem(g, c2, lane, j);
            *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(v[0] | (unsigned)v[1] << 16, v[2] | (unsigned)v[3] << 16,
                                                                    v[4] | (unsigned)v[5] << 16, v[6] | (unsigned)v[7] << 16);
        } else {
            const float* wout = g + (size_t)DECB_FRAGS * 256;
            float* ow = reinterpret_cast<float*>(dst);
            for (int i = lane; i < 4 * CD; i += 64) ow[i] = dect_f(dect_bf(wout[i]));
        }
    }
}


// partial weight-gradient tiles of the fused backward launches of one step, waiting for dect_reduce_kernel (ONE definition for
// giga_decoder_train16.hip and its caller giga_capi.hip)
struct DectPending { const float* partial[NHEADS]; int nwg[NHEADS]; int head_id[NHEADS]; int n; };

}  // namespace giga
