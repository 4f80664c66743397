// Issue cost of the VALU instructions the f16-class decoders live on (relu + fp32 -> f16 conversion of every hidden
// activation: 176 of them per 32-point tile and head), alone and beside the f16 MFMA of a sibling wave on the same SIMD.
//   part 1: cycles per instruction of one kind, 1 / 2 / 3 waves per SIMD (4 independent register chains per wave)
//   part 2: one wave per SIMD issues dependent-free v_mfma_f32_32x32x16_f16, a second wave per SIMD the VALU kind: elapsed
//           clocks of each alone and of both together (do they overlap?)
//   hipcc -O2 --offload-arch=gfx950 tools/valu_f16_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__device__ __forceinline__ void valu64(unsigned& a, unsigned& b, unsigned& c, unsigned& d, float x, float y) {
    // 64 instructions, 4 independent destination chains
    if constexpr (KIND == 0) asm volatile(REP16("v_cvt_pk_f16_f32 %0, %4, %5\nv_cvt_pk_f16_f32 %1, %4, %5\nv_cvt_pk_f16_f32 %2, %5, %4\nv_cvt_pk_f16_f32 %3, %5, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 1) asm volatile(REP16("v_pk_max_f16 %0, %0, 0\nv_pk_max_f16 %1, %1, 0\nv_pk_max_f16 %2, %2, 0\nv_pk_max_f16 %3, %3, 0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 2) asm volatile(REP16("v_max_f32 %0, %0, %4\nv_max_f32 %1, %1, %4\nv_max_f32 %2, %2, %5\nv_max_f32 %3, %3, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 3) asm volatile(REP16("v_cvt_f16_f32 %0, %4\nv_cvt_f16_f32 %1, %4\nv_cvt_f16_f32 %2, %5\nv_cvt_f16_f32 %3, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 4) asm volatile(REP16("v_cvt_pkrtz_f16_f32 %0, %4, %5\nv_cvt_pkrtz_f16_f32 %1, %4, %5\nv_cvt_pkrtz_f16_f32 %2, %5, %4\nv_cvt_pkrtz_f16_f32 %3, %5, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 5) asm volatile(REP16("v_max_i32 %0, 0, %0\nv_max_i32 %1, 0, %1\nv_max_i32 %2, 0, %2\nv_max_i32 %3, 0, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 6) asm volatile(REP16("v_add_f32 %0, %0, %4\nv_add_f32 %1, %1, %4\nv_add_f32 %2, %2, %5\nv_add_f32 %3, %3, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 7) asm volatile(REP16("v_cvt_f32_f16 %0, %4\nv_cvt_f32_f16 %1, %4\nv_cvt_f32_f16 %2, %5\nv_cvt_f32_f16 %3, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 8) asm volatile(REP16("v_fma_mixlo_f16 %0, %4, %5, %4 op_sel:[0,0,0] op_sel_hi:[0,0,0]\nv_fma_mixlo_f16 %1, %4, %5, %4 op_sel:[0,0,0] op_sel_hi:[0,0,0]\nv_fma_mixhi_f16 %2, %5, %4, %5 op_sel:[0,0,0] op_sel_hi:[0,0,0]\nv_fma_mixhi_f16 %3, %5, %4, %5 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 9) asm volatile(REP16("v_perm_b32 %0, %0, %1, %2\nv_perm_b32 %1, %1, %2, %3\nv_perm_b32 %2, %2, %3, %0\nv_perm_b32 %3, %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 10) asm volatile(REP16("v_pk_add_f16 %0, %0, %1\nv_pk_add_f16 %1, %1, %2\nv_pk_add_f16 %2, %2, %3\nv_pk_add_f16 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    if constexpr (KIND == 11) asm volatile(REP16("v_med3_f32 %0, %0, %4, %5\nv_med3_f32 %1, %1, %4, %5\nv_med3_f32 %2, %2, %4, %5\nv_med3_f32 %3, %3, %4, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
}
static const char* NAMES[] = {"v_cvt_pk_f16_f32", "v_pk_max_f16", "v_max_f32", "v_cvt_f16_f32", "v_cvt_pkrtz_f16_f32", "v_max_i32", "v_add_f32",
                              "v_cvt_f32_f16", "v_fma_mixlo/hi_f16", "v_perm_b32", "v_pk_add_f16", "v_med3_f32"};

template <int KIND>
__global__ void valu_only(long long* cyc, unsigned* sink, int iters) {
    unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    float x = threadIdx.x * 0.001f, y = 1.5f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) valu64<KIND>(a, b, c, d, x, y);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a + b + c + d == 0x12345678u) sink[0] = a;
}

// waves 0..3 (one per SIMD): MFMA stream; waves 4..7: VALU stream (mode 1 = mfma only, 2 = valu only, 3 = both)
template <int KIND>
__global__ __launch_bounds__(512) void beside_mfma(long long* cyc, unsigned* sink, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    float x = threadIdx.x * 0.001f, y = 1.5f;
    half8 A = {1, 2, 3, 4, 5, 6, 7, 8}, B = {1, 1, 1, 1, 1, 1, 1, 1};
    f32x16 d0, d1, d2, d3;
    for (int i = 0; i < 16; ++i) d0[i] = d1[i] = d2[i] = d3[i] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1)
            for (int i = 0; i < iters; ++i) {                  // 16 MFMAs per iteration: 512 clocks of matrix pipe
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, d2, 0, 0, 0);
                    d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, d3, 0, 0, 0);
                }
            }
    } else if (mode & 2) {
        for (int i = 0; i < iters; ++i) valu64<KIND>(a, b, c, d, x, y);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    if (a + b + c + d == 0x12345678u || d0[0] + d1[1] + d2[2] + d3[3] == 12345.f) sink[0] = a;
}

template <int KIND>
void run(long long* dc, unsigned* ds) {
    long long h[16];
    const int iters = 256;
    printf("%-22s", NAMES[KIND]);
    for (int waves = 4; waves <= 12; waves += 4) {             // 1, 2, 3 waves per SIMD
        hipLaunchKernelGGL(valu_only<KIND>, dim3(1), dim3(waves * 64), 0, 0, dc, ds, iters);
        hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
        printf("  %d/SIMD: %5.2f clk/instr/wave (%5.2f per SIMD slot)", waves / 4, (double)mx / (iters * 64), (double)mx / (iters * 64) / (waves / 4));
    }
    long long t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL(beside_mfma<KIND>, dim3(1), dim3(512), 0, 0, dc, ds, iters, mode);
        hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
        t[mode] = mx;
    }
    printf("  | beside f16 MFMA (16 MFMA : 64 VALU per iteration): mfma alone %lld, valu alone %lld, both %lld clk\n", t[1], t[2], t[3]);
}

int main() {
    long long* dc; unsigned* ds;
    hipMalloc(&dc, 16 * sizeof(long long)); hipMalloc(&ds, 64);
    run<0>(dc, ds); run<1>(dc, ds); run<2>(dc, ds); run<3>(dc, ds); run<4>(dc, ds); run<5>(dc, ds); run<6>(dc, ds); run<7>(dc, ds);
    run<8>(dc, ds); run<9>(dc, ds); run<10>(dc, ds); run<11>(dc, ds);
    return 0;
}
