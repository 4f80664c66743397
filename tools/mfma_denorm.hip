// Probe: does the f16 MFMA keep subnormal f16 A/B inputs?  (decides whether the split-operand decoder may store the
// low parts of weights/activations unscaled: |w_lo| ~ 2^-11 |w| is subnormal in f16 for |w| < 2^-3.)
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_denorm.hip -o giga_amd/lib/mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out, float a_val, float b_val) {
    const int lane = threadIdx.x;
    half8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }     // k-slot (hi=0, j=0) only
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    half8 a2 = {0, 0, 0, 0, 0, 0, 0, 0}, b2 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 16) { a2[0] = (_Float16)a_val; b2[0] = (_Float16)b_val; }
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, d, 0, 0, 0);
    if (lane == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)(_Float16)a_val; }
}

int main() {
    float* d; float h[3];
    hipMalloc(&d, sizeof(h));
    const float vals[][2] = {{3.0e-5f, 1024.f}, {1024.f, 3.0e-5f}, {6.0e-8f, 16384.f}, {3.0e-5f, 3.0e-5f}, {0.5f, 0.25f}};
    for (auto& v : vals) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, v[0], v[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const double exact = (double)(float)(_Float16)v[0] * (double)(float)(_Float16)v[1];
        printf("a=%g b=%g  f16(a)=%g  mfma32x32x16=%.9g  mfma16x16x32=%.9g  exact=%.9g  %s\n", v[0], v[1], h[2], h[0], h[1], exact,
               (h[0] == (float)exact && h[1] == (float)exact) ? "KEPT" : "FLUSHED/DIFF");
    }
    return 0;
}
