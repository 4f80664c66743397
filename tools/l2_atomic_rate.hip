// How fast can the workgroups of one XCD accumulate fp32 values into a SHARED image in their XCD's L2 with atomics that return nothing
// (global_atomic_add_f32 at workgroup scope: performed in the L2 the XCD owns)?  The question behind it: the 3x3 weight-gradient
// kernels write 128-256 private partial images per layer (290 MB per training step, read again by a reduce kernel); 8 XCD-local
// images filled by atomics would remove that traffic if the L2's atomic units keep up.
// 256 workgroups of 512 threads (one per CU), XCD found by XCC_ID; each adds ROUNDS x its 36-KiB tile (9216 floats = one 32 x 32
// block x 9 taps) into image [xcd][blk], blk = 0..NB-1 chosen per round -- all 32 workgroups of an XCD hit the same NB images.
//   hipcc -O2 --offload-arch=gfx950 tools/l2_atomic_rate.hip -o /tmp/l2a && /tmp/l2a
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NWG = 256, NT = 512, TILE = 9216, ROUNDS = 64;

template <int MODE>   // 0: atomic add, no return, workgroup scope; 1: the same at agent scope; 2: plain stores to a PRIVATE partial image (today's scheme)
__global__ __launch_bounds__(NT) void k(float* img, float* priv, long long* clocks, int nb) {
    const int xcd = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7;
    const float v = 1.0f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < ROUNDS; ++r) {
        const int blk = (r + blockIdx.x) % nb;
        float* dst = MODE == 2 ? priv + ((size_t)blockIdx.x * nb + blk) * TILE : img + ((size_t)xcd * nb + blk) * TILE;
        for (int e = threadIdx.x; e < TILE; e += NT) {
            if (MODE == 0) __hip_atomic_fetch_add(dst + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) __hip_atomic_fetch_add(dst + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dst[e] = v + (float)r;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* what, float* img, float* priv, long long* clocks, int nb) {
    (void)hipMemset(img, 0, (size_t)8 * 16 * TILE * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(NWG), dim3(NT), 0, 0, img, priv, clocks, nb);       // warm-up
    (void)hipMemset(img, 0, (size_t)8 * 16 * TILE * 4);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(NWG), dim3(NT), 0, 0, img, priv, clocks, nb);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h((size_t)8 * 16 * TILE);
    (void)hipMemcpy(h.data(), img, h.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (float x : h) sum += x;
    const double lanes = (double)NWG * ROUNDS * TILE;
    printf("%-64s images/XCD %2d: %8.1f us, %6.1f G lane-adds/s = %5.1f lanes/clk/XCD at 2.4 GHz, %6.2f TB/s of values; sum of the images %.0f (expected %.0f)\n",
           what, nb, ms * 1e3, lanes / (ms * 1e-3) * 1e-9, lanes / (ms * 1e-3) / 8 / 2.4e9, lanes * 4 / (ms * 1e-3) * 1e-12, sum, MODE == 2 ? 0.0 : lanes);
}

int main() {
    float *img, *priv; long long* clocks;
    (void)hipMalloc(&img, (size_t)8 * 16 * TILE * 4);
    (void)hipMalloc(&priv, (size_t)NWG * 16 * TILE * 4);
    (void)hipMalloc(&clocks, NWG * 8);
    for (int nb : {1, 4, 8, 16}) {
        run<0>("atomic add f32, no return, workgroup scope (XCD-local image)", img, priv, clocks, nb);
        run<1>("atomic add f32, no return, agent scope", img, priv, clocks, nb);
        run<2>("plain stores to a private partial image per workgroup", img, priv, clocks, nb);
    }
    return 0;
}
