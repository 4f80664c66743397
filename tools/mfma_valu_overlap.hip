// Does a fp32-input MFMA (v_mfma_f32_16x16x4_f32) of one wave co-execute with plain VALU work of ANOTHER wave on the same SIMD?
// And with the XDL f16 MFMA?  8 waves per workgroup (2 per SIMD), one workgroup per CU: waves 0-3 issue NM MFMAs, waves 4-7
// issue NV v_fma_f32 (4 independent chains each).  Prints the kernel time for (MFMA only), (VALU only), (both).
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv, int same) {
    const int wave = threadIdx.x >> 6;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    half8 ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(a + e); hb[e] = (_Float16)(1 + e); }
    const bool do_m = same || wave < 4, do_v = same || wave >= 4;
    if (do_m && !same) {
        for (int i = 0; i < nm; ++i) {
            if constexpr (KIND == 0) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d3, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d3, 0, 0, 0);
            }
        }
    }
    if (do_v && !same) {
        for (int i = 0; i < nv; ++i) {
            asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(b));
        }
    }
    if (same) {   // one wave interleaves: 4 MFMAs then nv/nm * 4 FMAs
        const int per = nm ? nv / nm : 0;
        for (int i = 0; i < nm; ++i) {
            if constexpr (KIND == 0) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d3, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d3, 0, 0, 0);
            }
            for (int q = 0; q < per; ++q)
                asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(b));
        }
    }
    f32x4 d = d0 + d1 + d2 + d3;
    out[blockIdx.x * 512 + threadIdx.x] = d[0] + d[1] + d[2] + d[3] + v0 + v1 + v2 + v3;
}

template <int KIND>
float run(float* out, int nm, int nv, int same) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, nm, nv, same);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, nm, nv, same);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int NM = 20000;                       // x4 MFMAs per wave
    for (int kind = 0; kind < 2; ++kind) {
        auto R = [&](int nm, int nv, int same) { return kind == 0 ? run<0>(out, nm, nv, same) : run<1>(out, nm, nv, same); };
        const char* nmn = kind == 0 ? "v_mfma_f32_16x16x4_f32 (32 cycles)" : "v_mfma_f32_16x16x32_f16 (XDL, 16 cycles at spec)";
        printf("== %s\n", nmn);
        const int NV = kind == 0 ? NM * 4 : NM * 2;   // VALU work worth about half the MFMA time
        printf("two waves per SIMD, one MFMA wave + one VALU wave:  MFMA only %8.1f us   VALU only %8.1f us   both %8.1f us\n",
               R(NM, 0, 0), R(0, NV, 0), R(NM, NV, 0));
        printf("every wave interleaves 4 MFMA + %d x4 FMA:          MFMA only %8.1f us   both %8.1f us\n", NV / NM, R(NM, 0, 1), R(NM, NV, 1));
    }
    return 0;
}
