// Can the 32 workgroups that share an XCD (and its L2) synchronise and exchange data through that L2 alone -- no agent-scope
// release / acquire (L2 write-back + invalidate, ~7 us) and no device-scope atomics (resolved outside the XCD, ~8 us for 256
// arrivals)?  256 workgroups of 768 threads, one per CU.  Each workgroup reads its XCD from the hardware register XCC_ID,
// takes a rank on that XCD, and then runs ROUNDS rounds of: write a 1-KiB record (round stamp), barrier on the XCD's counter
// (workgroup-scope atomic add = performed in the XCD's L2; spin on an L2 load), read the records of ALL workgroups of the
// XCD and check their stamps (every round uses a fresh region, as every U-Net layer writes a fresh buffer: a reader's L1
// cannot hold an older copy of a line it has never read).  Prints the blockIdx -> XCD map check, workgroups per XCD, stale reads, us per barrier.
//   hipcc -O2 --offload-arch=gfx950 tools/xcd_barrier.hip -o /tmp/xcdb && /tmp/xcdb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NWG = 256, NT = 768, ROUNDS = 200;

__global__ __launch_bounds__(NT) void k(unsigned* cnt /* [8][32] counters, zeroed */, unsigned* tickets /* [8][32] */,
                                        unsigned* rec /* [ROUNDS + 1][8][32][256]: a fresh region per round, as a U-Net layer writes a fresh buffer */, int* xcd_of /* [NWG] */, unsigned* stale, int device_scope) {
    __shared__ int s_rank;
    const int xcd = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;        // HW_REG_XCC_ID[3:0]
    if (threadIdx.x == 0) {
        xcd_of[blockIdx.x] = xcd;
        s_rank = (int)__hip_atomic_fetch_add(tickets + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const int rank = s_rank;
    unsigned* c = cnt + xcd * 32;
    // number of workgroups on this XCD: wait until the ticket counter stops at 32 (the launch gives every XCD 32)
    unsigned bad = 0;
    for (int r = 1; r <= ROUNDS; ++r) {
        unsigned* region = rec + (size_t)r * 8 * 32 * 256 + (size_t)xcd * 32 * 256;
        if (threadIdx.x < 256) region[rank * 256 + threadIdx.x] = (unsigned)r * 1000u + rank;   // the "layer output"
        __builtin_amdgcn_s_waitcnt(0x0F70);                                             // vmcnt(0): my stores are in the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            if (device_scope) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u * r) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned spins = 0;
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u * r) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24)) { atomicAdd(stale + 1, 1u); break; }
                }
            }
        }
        __syncthreads();
        // read every record of this XCD (plain loads: none of these lines can be in my L1 with this round's address+value)
        for (int w = threadIdx.x >> 8; w < 32; w += NT / 256) {
            const unsigned v = region[w * 256 + (threadIdx.x & 255)];
            if (v != (unsigned)r * 1000u + w) ++bad;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(stale, bad);
}

int main() {
    unsigned *cnt, *tickets, *rec, *stale; int* xcd_of;
    (void)hipMalloc(&cnt, 8 * 32 * 4); (void)hipMalloc(&tickets, 8 * 32 * 4); (void)hipMalloc(&rec, (size_t)(ROUNDS + 1) * 8 * 32 * 256 * 4);
    (void)hipMalloc(&stale, 8); (void)hipMalloc(&xcd_of, NWG * 4);
    for (int mode = 0; mode < 2; ++mode) {
        (void)hipMemset(cnt, 0, 8 * 32 * 4); (void)hipMemset(tickets, 0, 8 * 32 * 4); (void)hipMemset(stale, 0, 8);
        (void)hipMemset(rec, 0, (size_t)(ROUNDS + 1) * 8 * 32 * 256 * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(NWG), dim3(NT), 0, 0, cnt, tickets, rec, xcd_of, stale, mode);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int h[NWG]; unsigned st[2], tk[8 * 32];
        (void)hipMemcpy(h, xcd_of, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(st, stale, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(tk, tickets, sizeof tk, hipMemcpyDeviceToHost);
        int mism = 0; for (int i = 0; i < NWG; ++i) mism += h[i] != i % 8;
        printf("%s: %.1f us per round (write + barrier + read of 32 KiB); blockIdx %% 8 != XCC_ID for %d of %d workgroups; "
               "workgroups per XCD:", mode ? "agent-scope fences + device atomics" : "XCD-local (L2 only)          ", ms * 1e3 / ROUNDS, mism, NWG);
        for (int x = 0; x < 8; ++x) printf(" %u", tk[x * 32]);
        printf("; stale reads %u, spin timeouts %u\n", st[0], st[1]);
    }
    return 0;
}
