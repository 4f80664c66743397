// Device-side helpers shared by the gfx950 kernels.  CDNA4 only: 64-wide waves, MFMA, LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "giga_layout.h"
#include "giga_launch.h"

namespace giga {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = A(32 x 16, f16) * B(16 x 32, f16) + C.  lane (n = lane&31, hi = lane>>5) holds k-slots
// (hi, 0..7) of row n of A / column n of B; D: column lane&31, rows (r&3)+8*(r>>2)+4*hi.
__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// D = A(32 x 2, f32) * B(2 x 32, f32) + C, exact fp32 (k-ordered fma chain); lane holds k-slot hi.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 16x16 MFMA shapes: D row = 4*(lane>>4) + reg, column = lane&15; A[i=lane&15][k-slot lane>>4], B[k-slot][j=lane&15]
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4v mfma32_16(float a, float b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mfma16_16(half8 a, half8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ReLU as ONE instruction: a signed-integer max on the float's bits (negative floats, -0 included, are negative
// integers).  fmaxf(x, 0) on an MFMA result compiles to TWO v_max_f32: IEEE fmaxnum semantics put a canonicalising
// `v_max_f32 x, x` in front because the compiler cannot prove the accumulator is not a signalling NaN, and with four
// MFMA-result ReLUs per fp32 MFMA step the extra VALU slot is not free (fp32 MFMA and VALU issue serially).  (An
// inline-asm v_max_f32 is not an option: the hazard recogniser does not see into it and the MFMA -> VALU wait states
// go missing.)  Bit-identical to fmaxf for every non-NaN input; NaNs with the sign bit set become 0, others pass.
__device__ __forceinline__ float relu(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// fmaxf form.  The backward decoder keeps it: with the integer form its register allocation tips over
// (462 -> 512 VGPRs + 16 spills).
__device__ __forceinline__ float relu_ieee(float x) { return __builtin_fmaxf(x, 0.f); }

// round-to-nearest f16 then relu (== relu then round) of D registers 8c..8c+7 -> B operand of the next
// layer's chunk c.  4x v_cvt_pk_f16_f32 + 4x v_pk_max_f16 instead of 16 canonicalising v_max_f32.
__device__ __forceinline__ half8 pack_relu8(const f32x16& d, int c) {
    half8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (half_t)d[8 * c + j];
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_elementwise_max(x, z);
}

// f16x3 split of relu(D registers 8c..8c+7): hi = f16(x) (round to nearest), lo = f16(x - hi); hi + lo carries x to
// ~2^-22 relative.  The subtraction is written as an fma on the widened half so that it selects v_fma_mix_f32 (one
// instruction, exact: x - hi has at most 13 significant bits).
// global_store_dword in its saddr form: uniform 64-bit base in SGPRs + 32-bit per-lane byte offset.  (The compiler hoists the
// zero-extension of the offset out of loops and then selects a 64-bit VALU add per store instead.)
__device__ __forceinline__ void store_f32_saddr(float* uniform_base, unsigned byte_off, float v) {
    asm volatile("global_store_dword %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(uniform_base) : "memory");
}

__device__ __forceinline__ void store_f16_saddr(_Float16* uniform_base, unsigned byte_off, _Float16 v) {
    asm volatile("global_store_short %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(uniform_base) : "memory");
}
__device__ __forceinline__ void store_saddr(float* b, unsigned o, float v) { store_f32_saddr(b, o, v); }
__device__ __forceinline__ void store_saddr(_Float16* b, unsigned o, _Float16 v) { store_f16_saddr(b, o, v); }

// v_permlane32_swap / v_permlane16_swap (gfx950): a's upper half (odd 16-lane rows) <-> b's lower half (even rows); plain VALU,
// no LDS.  Inline asm: the builtin's second result is mis-assigned by this compiler (both results alias the first operand).
// The leading s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see through the asm.
__device__ __forceinline__ void lane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane16_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// f16x3 split of relu(D registers 8c..8c+7): hi = f16(x) (round to nearest), lo = f16(x - hi); hi + lo carries x to ~2^-22
// relative.  Compiles to v_cvt_pk_f16_f32 (hi pairs) + v_cvt_f32_f16 / v_sub_f32 / v_cvt_f16_f32 per element (~4 VALU per
// element).  MEASURED DEAD END (round 3): the two-instruction form with v_fma_mixlo/hi_f16 in inline asm (2.5 VALU per element)
// is only 1.5-5 % faster on the f16x3 decoders (they wait on barriers and latency, not on VALU issue) and is unsafe without a
// hand-placed `s_nop 1`: the compiler keeps two wait states between a VALU write and an MFMA that reads the register, and does
// not see through the asm -- the separable-fc_c decoder read stale lo halves (6e-5 errors; profiles/r03_notes).
__device__ __forceinline__ void split_relu8(const f32x16& d, int c, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = relu(d[8 * c + j]);
        const half_t h = (half_t)x;
        hi[j] = h;
        lo[j] = (half_t)__builtin_fmaf((float)h, -1.0f, x);
    }
}
__device__ __forceinline__ void split8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const half_t h = (half_t)x[j];
        hi[j] = h;
        lo[j] = (half_t)__builtin_fmaf((float)h, -1.0f, x[j]);
    }
}

// ---- coordinates: reference ConvONets/common.py:238-261 with padding = 0 -------------------------
// xy = p / (1 + 0 + 10e-6) + 0.5 ; >= 1 -> 1 - 10e-6 ; < 0 -> 0     (fp32, true division)
__device__ __forceinline__ float norm_coord(float p) {
    float v = p / 1.00001f + 0.5f;
    v = v >= 1.0f ? 0.99999f : v;
    v = v < 0.0f ? 0.0f : v;
    return v;
}
// pixel coordinate of F.grid_sample(align_corners=True, padding_mode='border') on a 40-wide axis,
// from the normalised coordinate: vgrid = 2*xy - 1 ; ix = ((vgrid + 1) / 2) * 39 ; clip to [0, 39]
__device__ __forceinline__ float pix_coord(float xy) {
    float g = 2.0f * xy - 1.0f;
    float ix = ((g + 1.0f) * 0.5f) * 39.0f;
    return fminf(fmaxf(ix, 0.0f), 39.0f);
}

// ---- cheap index arithmetic (the decoders run it per lane per round; hardware has no integer divide) ----
// scene of point g:  b = g / N, r = g % N   via fp32 reciprocal + one fix-up step (exact for g < 2^31)
__device__ __forceinline__ void split_scene(long long g, int N, float invN, int& b, int& r) {
    int q = (int)((float)g * invN);
    long long rem = g - (long long)q * N;
    if (rem < 0) { --q; rem += N; }
    if (rem >= N) { ++q; rem -= N; }
    b = q; r = (int)rem;
}
// n / d for n * (d-1) < 2^32 with m = ceil(2^32 / d) computed on the host
__device__ __forceinline__ int div_magic(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }

struct Bilin {          // 4-tap footprint on a 40x40 plane, element offsets in pixels
    int o00, o01, o10, o11;
    float w00, w01, w10, w11;
};
// u indexes W (first listed plane axis), v indexes H (second)  -- decoder.py:117-122
__device__ __forceinline__ Bilin bilin_setup(float u, float v) {
    float fx = pix_coord(u), fy = pix_coord(v);
    float x0f = floorf(fx), y0f = floorf(fy);
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = min(x0 + 1, RES - 1), y1 = min(y0 + 1, RES - 1);
    float ax = fx - x0f, ay = fy - y0f;      // weights as aten grid_sampler_2d computes them
    float bx = (x0f + 1.0f) - fx, by = (y0f + 1.0f) - fy;
    Bilin b;
    b.o00 = y0 * RES + x0; b.o01 = y0 * RES + x1; b.o10 = y1 * RES + x0; b.o11 = y1 * RES + x1;
    b.w00 = bx * by; b.w01 = ax * by; b.w10 = bx * ay; b.w11 = ax * ay;
    if (x0 + 1 > RES - 1) b.w01 = b.w11 = 0.f;   // out-of-bounds taps contribute zero (never hit: xy <= 0.99999)
    if (y0 + 1 > RES - 1) b.w10 = b.w11 = 0.f;
    return b;
}


// XCD-aware work mapping (speed only, never correctness): workgroup i runs on XCD i % 8 (observed dispatch order), and
// each XCD has its own 4 MiB L2.  Consecutive work items share data (the points of one scene sample the same planes),
// so item indices are permuted such that XCD x gets the x-th contiguous eighth of the items instead of every eighth
// item: an XCD then touches 1/8 of the scenes and its L2 keeps their planes resident.
__device__ __forceinline__ int xcd_swizzle(int i, int n) {
    const int m = n & ~7;                    // the largest multiple of 8; the remainder keeps its place
    return i < m ? (i & 7) * (m >> 3) + (i >> 3) : i;
}

}  // namespace giga
