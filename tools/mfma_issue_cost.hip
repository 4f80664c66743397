// What does ONE extra instruction cost when it is interleaved 1:1 with fp32-input MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles each)
// in the SAME wave?  Per loop iteration: 16 MFMAs (4 independent accumulator chains) each followed by one instruction of the
// kind under test.  One workgroup per CU; WAVES = 4 (one wave per SIMD) or 8 (two per SIMD).  Prints clocks per iteration of the whole workgroup
// (first wave entry to last wave exit: the SIMD arbiter favours the older wave, a single wave's time says nothing).
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define M(d) "v_mfma_f32_16x16x4_f32 %" #d ", %4, %5, %" #d "\n"
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    __shared__ float sm[4096];
    sm[threadIdx.x] = threadIdx.x; sm[threadIdx.x + 512] = 1.f;
    __syncthreads();
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    f32x2 p0 = {a, a}, p1 = {b, b}, p2 = p0, p3 = p1;
    unsigned la = (threadIdx.x & 255) * 4, la4 = (threadIdx.x & 63) * 16;
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0)
            asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
#define X1 "v_add_f32 %6, %6, %5\n"
#define X2 "v_add_f32 %7, %7, %5\n"
#define X3 "v_add_f32 %8, %8, %5\n"
#define X4 "v_add_f32 %9, %9, %5\n"
        if constexpr (KIND == 1)
            asm volatile(M(0) X1 M(1) X2 M(2) X3 M(3) X4 M(0) X1 M(1) X2 M(2) X3 M(3) X4 M(0) X1 M(1) X2 M(2) X3 M(3) X4 M(0) X1 M(1) X2 M(2) X3 M(3) X4
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#define P1 "v_pk_add_f32 %6, %6, %7\n"
#define P2 "v_pk_add_f32 %8, %8, %9\n"
        if constexpr (KIND == 2)
            asm volatile(M(0) P1 M(1) P2 M(2) P1 M(3) P2 M(0) P1 M(1) P2 M(2) P1 M(3) P2 M(0) P1 M(1) P2 M(2) P1 M(3) P2 M(0) P1 M(1) P2 M(2) P1 M(3) P2
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(p0), "v"(p1), "v"(p2), "v"(p3));
#define ML(d) "v_mfma_f32_16x16x4_f32 %" #d ", %8, %9, %" #d "\n"
#define L1 "ds_read_b32 %4, %10\n"
#define L2 "ds_read_b32 %5, %10 offset:1024\n"
#define L3 "ds_read_b32 %6, %10 offset:2048\n"
#define L4 "ds_read_b32 %7, %10 offset:3072\n"
        if constexpr (KIND == 3)
            asm volatile(ML(0) L1 ML(1) L2 ML(2) L3 ML(3) L4 ML(0) L1 ML(1) L2 ML(2) L3 ML(3) L4 ML(0) L1 ML(1) L2 ML(2) L3 ML(3) L4 ML(0) L1 ML(1) L2 ML(2) L3 ML(3) L4
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3) : "v"(a), "v"(b), "v"(la));
#define N "s_nop 0\n"
        if constexpr (KIND == 4)
            asm volatile(M(0) N M(1) N M(2) N M(3) N M(0) N M(1) N M(2) N M(3) N M(0) N M(1) N M(2) N M(3) N M(0) N M(1) N M(2) N M(3) N
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
#define S "s_add_u32 s20, s20, 1\n"
        if constexpr (KIND == 5)
            asm volatile(M(0) S M(1) S M(2) S M(3) S M(0) S M(1) S M(2) S M(3) S M(0) S M(1) S M(2) S M(3) S M(0) S M(1) S M(2) S M(3) S
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "s20", "scc");
#define Y "v_max_i32 %6, 0, %6\n"
        if constexpr (KIND == 6)   // four VALU per MFMA
            asm volatile(M(0) X1 X2 X3 X4 M(1) X1 X2 X3 X4 M(2) X1 X2 X3 X4 M(3) X1 X2 X3 X4 M(0) X1 X2 X3 X4 M(1) X1 X2 X3 X4 M(2) X1 X2 X3 X4 M(3) X1 X2 X3 X4
                         M(0) X1 X2 X3 X4 M(1) X1 X2 X3 X4 M(2) X1 X2 X3 X4 M(3) X1 X2 X3 X4 M(0) X1 X2 X3 X4 M(1) X1 X2 X3 X4 M(2) X1 X2 X3 X4 M(3) X1 X2 X3 X4
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#define MQ(d) "v_mfma_f32_16x16x4_f32 %" #d ", %6, %7, %" #d "\n"
#define Q1 "ds_read_b128 %4, %8\n"
#define Q2 "ds_read_b128 %5, %8 offset:4096\n"
        if constexpr (KIND == 8) {  // one 1-KiB LDS read per MFMA
            f32x4 q0, q1;
            asm volatile(MQ(0) Q1 MQ(1) Q2 MQ(2) Q1 MQ(3) Q2 MQ(0) Q1 MQ(1) Q2 MQ(2) Q1 MQ(3) Q2 MQ(0) Q1 MQ(1) Q2 MQ(2) Q1 MQ(3) Q2 MQ(0) Q1 MQ(1) Q2 MQ(2) Q1 MQ(3) Q2
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(q0), "=&v"(q1) : "v"(a), "v"(b), "v"(la4));
            l0 += q0[0] + q1[0];
        }
        if constexpr (KIND == 9) {  // three 1-KiB LDS reads per 8 MFMAs (the conv16 fp32 ratio, NB = 2)
            f32x4 q0, q1;
            asm volatile(MQ(0) Q1 MQ(1) Q2 MQ(2) Q1 MQ(3) MQ(0) MQ(1) MQ(2) MQ(3) MQ(0) Q1 MQ(1) Q2 MQ(2) Q1 MQ(3) MQ(0) MQ(1) MQ(2) MQ(3)
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(q0), "=&v"(q1) : "v"(a), "v"(b), "v"(la4));
            l0 += q0[0] + q1[0];
        }
        if constexpr (KIND == 7)   // VALU only: 64 adds
            asm volatile(X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4
                         X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4 X1 X2 X3 X4
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { cyc[threadIdx.x >> 6] = t1 - t0; cyc[8 + (threadIdx.x >> 6)] = t0; cyc[16 + (threadIdx.x >> 6)] = t1; }
    f32x4 d = d0 + d1 + d2 + d3;
    out[blockIdx.x * 512 + threadIdx.x] = d[0] + d[1] + d[2] + d[3] + v0 + v1 + v2 + v3 + p0[0] + p2[1] + l0 + l1 + l2 + l3;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc, int extra_per_mfma) {
    const int iters = 2000;
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), 0, 0, out, iters, cyc);
        (void)hipDeviceSynchronize();
        long long cc[24]; (void)hipMemcpy(cc, cyc, 24 * 8, hipMemcpyDeviceToHost);
        long long lo = cc[8], hi = cc[16];                      // first entry .. last exit over the workgroup's waves
        for (int w = 1; w < waves; ++w) { lo = cc[8 + w] < lo ? cc[8 + w] : lo; hi = cc[16 + w] > hi ? cc[16 + w] : hi; }
        const double per_iter = (double)(hi - lo) / iters;
        const int nm = KIND == 7 ? 0 : 16;
        printf("%-34s %d wave(s)/SIMD: %8.1f clocks per iteration (%2d MFMA + %2d other)", name, waves / 4, per_iter, nm, extra_per_mfma);
        if (nm) printf("  -> %6.2f clocks per extra instruction beyond %d x 32 (per wave)", (per_iter / (waves / 4) - nm * 32.0) / (extra_per_mfma ? extra_per_mfma : 1), nm);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 24 * 8);
    if (only < 0 || only == 0) run<0>("MFMA only", out, cyc, 0);
    if (only < 0 || only == 1) run<1>("+ v_add_f32 per MFMA", out, cyc, 16);
    if (only < 0 || only == 2) run<2>("+ v_pk_add_f32 per MFMA", out, cyc, 16);
    if (only < 0 || only == 3) run<3>("+ ds_read_b32 per MFMA", out, cyc, 16);
    if (only < 0 || only == 4) run<4>("+ s_nop 0 per MFMA", out, cyc, 16);
    if (only < 0 || only == 5) run<5>("+ s_add_u32 per MFMA", out, cyc, 16);
    if (only < 0 || only == 6) run<6>("+ 4 v_add_f32 per MFMA", out, cyc, 64);
    if (only < 0 || only == 7) run<7>("64 v_add_f32 only", out, cyc, 64);
    if (only < 0 || only == 8) run<8>("+ ds_read_b128 per MFMA", out, cyc, 16);
    if (only < 0 || only == 9) run<9>("+ 3 ds_read_b128 per 8 MFMA", out, cyc, 6);
    return 0;
}
