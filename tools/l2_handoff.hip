// How fast does a workgroup read what ANOTHER workgroup of its XCD has just written -- the layer-to-layer hand-off of the
// persistent U-Net kernels -- and does the cache policy of the stores / loads matter?  256 workgroups of 512 threads, one per
// CU; ranks by ticket per XCD (XCC_ID).  Per round: every workgroup writes a 64-KiB region (16 B per lane and store), an XCD-local
// barrier (workgroup-scope atomic in the L2, as the kernels use), then it reads the region of rank + 1 on the same XCD (8 loads of
// 16 B per lane in flight) and times that with s_memtime.  Variants: store / load cache policy bits (sc0, sc1, nt), the number of
// workgroups per XCD that read (all 32, or 4: latency rather than bandwidth), and a COLD read of a region nobody has touched.
//   hipcc -O2 --offload-arch=gfx950 tools/l2_handoff.hip -o /tmp/l2h && /tmp/l2h
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 512, ROUNDS = 24, REG_BYTES = 64 * 1024, IT = REG_BYTES / (NT * 16);   // 8 x 16 B per lane

template <int POL> __device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}
template <int POL> __device__ __forceinline__ u32x4 ld16(const u32x4* p) {
    u32x4 v;
    if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SP, int LP>
__global__ __launch_bounds__(NT) void k(unsigned* cnt, unsigned* tickets, u32x4* rec /* [ROUNDS][8][32][REG] */, const u32x4* cold,
                                        long long* clocks /* [2][NWG] */, unsigned* bad_out, int readers, int use_cold, int second, int warm, int self, int pre_stride = 0) {
    __shared__ int s_rank;
    const int xcd = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7;
    if (threadIdx.x == 0) s_rank = (int)__hip_atomic_fetch_add(tickets + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const int rank = s_rank;
    unsigned* c = cnt + xcd * 32;
    long long total = 0, total2 = 0;
    unsigned bad = 0;
    constexpr int RV = REG_BYTES / 16;
    for (int r = 0; r < ROUNDS; ++r) {
        // warm: the SAME two 16-MiB buffers every round (ping-pong), as a recycled activation workspace gives the real kernels:
        // TLB entries and L2 tags of the regions are warm after the first two rounds; self: the reader is the writer (L2-hit control)
        const size_t rr = warm ? (size_t)(r & 1) : (size_t)r;
        const int src_rank = self ? rank : (rank + 1) % 32;
        u32x4* mine = rec + ((rr * 8 + xcd) * 32 + rank) * RV;
        const u32x4* theirs = use_cold ? cold + ((rr * 8 + xcd) * 32 + rank) * RV
                                       : rec + ((rr * 8 + xcd) * 32 + src_rank) * RV;
        const unsigned stamp = (unsigned)r * 1000u + (unsigned)src_rank;
        if (pre_stride > 0) {
            // translation prefetch: BEFORE the producer has written it, touch one word per `pre_stride` bytes of the region this workgroup
            // will read after the barrier (the values are stale and discarded; the page-table walk is what is wanted)
            const int nt = REG_BYTES / pre_stride;
            if ((int)threadIdx.x < nt) {
                unsigned junk;
                asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(junk) : "v"(reinterpret_cast<const char*>(theirs) + (size_t)threadIdx.x * pre_stride) : "memory");
                bad += junk == 0xFFFFFFFEu;
            }
        }
        if (pre_stride == -2 || pre_stride == -3) {
            // the same with device-scope (sc1) loads, which must not leave a copy in this CU's vector L1; -3: issued by the CONSUMER
            // for the region it will read (the question: does its later plain read see stale data? -> "wrong values")
            const u32x4* tgt = pre_stride == -2 ? mine : theirs;
            u32x4 junk[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i) junk[i] = ld16<2>(tgt + i * NT + threadIdx.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < IT; ++i) asm volatile("" : "+v"(junk[i]));
#pragma unroll
            for (int i = 0; i < IT; ++i) bad += junk[i].w == 0xFFFFFFFEu;
            if (pre_stride == -3) __syncthreads();
        }
        if (pre_stride == -1) {
            // write-allocate by hand: the PRODUCER reads the whole region it is about to write (stale values, discarded), so that its
            // stores hit lines that are already in the L2
            u32x4 junk[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i) junk[i] = ld16<0>(mine + i * NT + threadIdx.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < IT; ++i) asm volatile("" : "+v"(junk[i]));    // ALL four registers of every load stay allocated until here:
#pragma unroll                                                                 // the compiler believes an asm load completes at once and
            for (int i = 0; i < IT; ++i) bad += junk[i].w == 0xFFFFFFFEu;      // re-uses the registers of unused components while it is in flight
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) st16<SP>(mine + i * NT + threadIdx.x, u32x4{(unsigned)r * 1000u + rank, threadIdx.x, (unsigned)i, 7u});
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u * (r + 1)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (rank < readers) {
            const long long t0 = __builtin_amdgcn_s_memtime();
            u32x4 v[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i) v[i] = ld16<LP>(theirs + i * NT + threadIdx.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < IT; ++i) asm volatile("" : "+v"(v[i]));
            const long long t1 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const long long t2 = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) total += t2 - t0;
            if (second) {                                  // a region that ANOTHER workgroup of this XCD has just read: an L2 hit if reads allocate
                const u32x4* again = rec + ((rr * 8 + xcd) * 32 + (rank + 2) % 32) * RV;
                __syncthreads();
                const long long t3 = __builtin_amdgcn_s_memtime();
                u32x4 w[IT];
#pragma unroll
                for (int i = 0; i < IT; ++i) w[i] = ld16<LP>(again + i * NT + threadIdx.x);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < IT; ++i) asm volatile("" : "+v"(w[i]));
                __syncthreads();
                const long long t4 = __builtin_amdgcn_s_memtime();
                if (threadIdx.x == 0) total2 += t4 - t3;
#pragma unroll
                for (int i = 0; i < IT; ++i) bad += w[i].z != (unsigned)i;
            }
            if (!use_cold)
#pragma unroll
                for (int i = 0; i < IT; ++i) bad += v[i].x != stamp || v[i].z != (unsigned)i;
            else bad += v[0].x == 0xFFFFFFFFu;
            (void)t1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { clocks[blockIdx.x] = rank < readers ? total : -1; clocks[NWG + blockIdx.x] = total2; }
    if (bad) atomicAdd(bad_out, bad);
}

template <int SP, int LP>
static void run(const char* what, unsigned* cnt, unsigned* tickets, u32x4* rec, u32x4* cold, long long* clocks, unsigned* bad, int readers, int use_cold, int second = 0, int warm = 0, int self = 0, int pre_stride = 0) {
    (void)hipMemset(cnt, 0, 8 * 32 * 4); (void)hipMemset(tickets, 0, 8 * 32 * 4); (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k<SP, LP>), dim3(NWG), dim3(NT), 0, 0, cnt, tickets, rec, cold, clocks, bad, readers, use_cold, second, warm, self, pre_stride);
    (void)hipDeviceSynchronize();
    long long h[2 * NWG]; unsigned b;
    (void)hipMemcpy(h, clocks, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0; long long mx = 0;
    for (int i = 0; i < NWG; ++i) if (h[i] >= 0) { sum += (double)h[i]; ++n; if (h[i] > mx) mx = h[i]; }
    const double per = sum / n / ROUNDS;
    printf("%-44s readers/XCD %2d: %7.0f clocks per 64-KiB read (slowest workgroup %7.0f) = %5.1f B/clk/CU, wrong values %u\n", what, readers, per,
           (double)mx / ROUNDS, REG_BYTES / per, b);
    fflush(stdout);
    if (second) {
        double s2 = 0; for (int i = 0; i < NWG; ++i) if (h[i] >= 0) s2 += (double)h[NWG + i];
        fflush(stdout);
        printf("%-44s                 second read (region another workgroup just read): %7.0f clocks = %5.1f B/clk/CU\n", "", s2 / n / ROUNDS, REG_BYTES / (s2 / n / ROUNDS));
    }
}

int main(int argc, char** argv) {
    const bool warm_only = argc > 1 && argv[1][0] == 'w';
    unsigned *cnt, *tickets, *bad; u32x4 *rec, *cold; long long* clocks;
    const size_t RB = (size_t)ROUNDS * 8 * 32 * REG_BYTES;      // 384 MiB
    (void)hipMalloc(&cnt, 8 * 32 * 4); (void)hipMalloc(&tickets, 8 * 32 * 4); (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&rec, RB); (void)hipMalloc(&cold, RB); (void)hipMalloc(&clocks, 2 * NWG * 8);
    (void)hipMemset(rec, 0, RB); (void)hipMemset(cold, 1, RB);
    if (warm_only) {
        // round 5: (i) warm regions, (ii) the second-read variant recorded, (iii) reader = writer as the L2-hit control
        for (int readers : {32, 4}) {
            run<0, 0>("WARM ping-pong: store plain / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 1, 0);
            run<2, 0>("WARM ping-pong: store sc1 / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 1, 0);
            run<0, 4>("WARM ping-pong: store plain / load nt", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 1, 0);
            run<0, 0>("WARM, reader = writer (L2-hit control)", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 1, 1);
            run<2, 0>("WARM, reader = writer, store sc1", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 1, 1);
            run<0, 0>("fresh regions, reader = writer", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 0, 1);
            run<0, 0>("fresh regions: store plain / load plain + 2nd", cnt, tickets, rec, cold, clocks, bad, readers, 0, 1, 0, 0);
            run<0, 0>("WARM ping-pong: store plain / load plain + 2nd", cnt, tickets, rec, cold, clocks, bad, readers, 0, 1, 1, 0);
            run<0, 0>("fresh + translation prefetch every 4 KiB", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 0, 0, 4096);
            run<0, 0>("fresh, producer READS its region before writing", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 0, 0, -1);
            run<0, 4>("fresh, producer reads first / consumer load nt", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 0, 0, -1);
            run<0, 0>("fresh, producer reads first with sc1 loads", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 0, 0, -2);
            run<0, 0>("fresh, CONSUMER pre-reads with sc1 loads", cnt, tickets, rec, cold, clocks, bad, readers, 0, 0, 0, 0, -3);
        }
        return 0;
    }
    for (int readers : {32, 4}) {
        run<0, 0>("store plain / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<0, 0>("store plain / load plain + second read", cnt, tickets, rec, cold, clocks, bad, readers, 0, 1);
        run<2, 0>("store sc1 / load plain + second read", cnt, tickets, rec, cold, clocks, bad, readers, 0, 1);
        run<0, 1>("store plain / load sc0", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<0, 2>("store plain / load sc1", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<0, 4>("store plain / load nt", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<1, 0>("store sc0 / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<2, 0>("store sc1 / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<3, 0>("store sc0 sc1 / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<4, 0>("store nt / load plain", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        run<4, 4>("store nt / load nt", cnt, tickets, rec, cold, clocks, bad, readers, 0);
        (void)hipMemset(cold, 1, RB);            // (refresh: evicts the caches' copies of `rec`; the cold region itself was last touched by this memset)
        run<0, 0>("COLD region (not written in this launch)", cnt, tickets, rec, cold, clocks, bad, readers, 1);
    }
    return 0;
}
