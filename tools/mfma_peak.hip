// MFMA and HBM peak micro-benchmark for gfx950 ("confirm peaks on the box with an MFMA micro-benchmark and record
// both spec and measured peak").  Every wave issues a long chain of independent MFMAs (4 accumulator sets, operands
// in registers, no memory traffic); 256 CUs x 4 SIMDs x 2 waves.  Prints achieved TFLOP/s per instruction shape and
// the implied clock (cycles per instruction are the documented pass counts: 16 passes = 64 cycles for the 32x32
// shapes, 8 passes = 32 cycles for the 16x16 shapes).
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_peak.hip -o giga_amd/lib/mfma_peak && giga_amd/lib/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Shape { F32_32x32x2, F32_16x16x4, F16_32x32x16, F16_16x16x32 };

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_chain(float* out, int iters, float seed) {
    const float a32 = seed + threadIdx.x * 1e-6f, b32 = seed * 0.5f;
    f16x8 a16, b16;
    for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)(a32 + i); b16[i] = (_Float16)(b32 - i); }
    float sink = 0.f;
    if constexpr (SHAPE == F32_32x32x2 || SHAPE == F16_32x32x16) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
            if constexpr (SHAPE == F32_32x32x2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a32, b32, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a32, b32, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a32, b32, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a32, b32, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, c3, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) sink += c0[i] + c1[i] + c2[i] + c3[i];
    } else {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
            if constexpr (SHAPE == F32_16x16x4) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, c3, 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) sink += c0[i] + c1[i] + c2[i] + c3[i];
    }
    if (sink == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = sink;   // keeps the chain alive
}

template <int SHAPE>
static void run(const char* name, double flop_per_inst, int cycles_per_inst, double spec_tflops, float* out, int ncu) {
    const int iters = 20000, blocks = ncu * 2;                   // 2 WGs x 4 waves per CU = 2 waves per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mfma_chain<SHAPE><<<blocks, 256>>>(out, iters / 10, 1.0f);  // warm-up (clock ramp)
    mfma_chain<SHAPE><<<blocks, 256>>>(out, iters, 1.0f);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        mfma_chain<SHAPE><<<blocks, 256>>>(out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts_per_simd = 2.0 * 4 * iters;              // 2 waves share one SIMD's matrix core
    const double total = (double)blocks * 4 * 4 * iters * flop_per_inst;
    const double tflops = total / (best * 1e-3) / 1e12;
    const double ghz = insts_per_simd * cycles_per_inst / (best * 1e-3) / 1e9;
    printf("%-28s %8.3f ms  %8.1f TFLOP/s  (spec %6.1f, %5.1f %%)  implied matrix-core clock %.2f GHz\n", name, best, tflops,
           spec_tflops, 100.0 * tflops / spec_tflops, ghz);
}

// HBM: streaming read (sum reduction so that nothing is elided) and copy over a buffer far larger than the 256 MB
// Infinity Cache; float4 per lane, grid-stride, 8 workgroups per CU.
__global__ __launch_bounds__(256) void hbm_read(const f32x4* __restrict__ src, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = __builtin_nontemporal_load(src + i);
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void hbm_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

static void run_hbm(int ncu) {
    const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(f32x4);
    f32x4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        double best = 1e30;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hbm_read<<<ncu * 8, 256>>>(a, n, out);
            else hbm_copy<<<ncu * 8, 256>>>(a, b, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double moved = mode == 0 ? (double)bytes : 2.0 * bytes;
        printf("%-28s %8.3f ms  %8.1f GB/s   (spec 8000, %5.1f %%)\n", mode == 0 ? "HBM read 4 GiB" : "HBM copy 4 GiB (rd+wr)", best,
               moved / (best * 1e-3) / 1e9, 100.0 * moved / (best * 1e-3) / 8e12);
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(out));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s  CUs %d  clockRate %.0f MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000.0);
    float* out;
    CK(hipMalloc(&out, 1 << 24));
    const int ncu = prop.multiProcessorCount;
    run<F32_32x32x2>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 64, 157.3, out, ncu);
    run<F32_16x16x4>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, 32, 157.3, out, ncu);
    run<F16_32x32x16>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 32, 2500.0, out, ncu);
    run<F16_16x16x32>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, 16, 2500.0, out, ncu);
    CK(hipFree(out));
    run_hbm(ncu);
    return 0;
}
