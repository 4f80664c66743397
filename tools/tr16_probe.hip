// ds_read_b64_tr_b16 (gfx950): which lane gets which 16-bit element, and what the decoder-backward tile layout costs.
//   hipcc -O2 --offload-arch=gfx950 tools/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
// Hypothesis H (cdna_hip_programming.md, T10): inside a 16-lane group the 16 supplied 8-byte pieces form a [4 rows][4 pieces]
// block (lane 4*row + piece supplies piece `piece` of row `row`); lane i of the group receives column i of that 4 x 16 block:
//   result[lane = 16 g + i][j] = u16 at (address supplied by lane 16 g + 4 j + (i >> 2)) + 2 * (i & 3).
// Part 1 checks H with contiguous and with randomly permuted per-lane addresses.  Part 2 checks the tile layout of
// csrc/giga_decoder_bwd16.hip: a [32 points][32 features] bf16 tile whose 8-byte piece (point p, feature group q) sits at
// 64 p + 8 (q ^ ((p >> 1) & 7)), written from the chain layout (lane = point) and read as an MFMA operand (lane = feature,
// k-slots = points).  Part 3 times reads / writes of that layout against the unswizzled and the 72-byte-row ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;

__device__ __forceinline__ v4s tr_read(const unsigned char* base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(base + byte_off));
}

__global__ void probe_kernel(const int* lane_addr /* [64] byte offsets */, short* out /* [64][4] */) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    short* e = reinterpret_cast<short*>(lds);
    for (int i = threadIdx.x; i < 4096; i += 64) e[i] = (short)i;
    __syncthreads();
    const v4s r = tr_read(lds, lane_addr[threadIdx.x]);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}

// tile layouts: byte address of piece (point p, feature group q = features 4q..4q+3)
template <int LAYOUT> __host__ __device__ inline int piece_addr(int p, int q) {
    if (LAYOUT == 0) return 64 * p + 8 * (q ^ ((p >> 1) & 7));      // swizzled, 64-byte rows
    if (LAYOUT == 1) return 64 * p + 8 * q;                           // natural
    if (LAYOUT == 3) return 64 * p + 8 * (q ^ ((p >> 2) & 7));      // swizzled by the row QUAD: rows p and p + 16 land 16 banks apart
    return 72 * p + 8 * q;                                            // padded rows
}
// write side: lane (n = point, hi) holds, for chunk c, features 16c + 4hi + {0..3} (piece 0) and 16c + 8 + 4hi + {0..3} (piece 1)
// read side:  lane (o = l & 31 feature, hk = l >> 5), chunk, t -> points 16 chunk + 8 hk + 4 t + {0..3}
template <int LAYOUT> __host__ __device__ inline int read_addr(int l, int chunk, int t) {
    const int p = 16 * chunk + 8 * (l >> 5) + 4 * t + ((l & 15) >> 2);
    const int q = 4 * ((l >> 4) & 1) + (l & 3);
    return piece_addr<LAYOUT>(p, q);
}

template <int LAYOUT>
__global__ void tile_kernel(short* out /* [64][16]: chunk, t, j */, long long* clocks /* [2]: write loop, read loop */, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 2560];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char* tile = lds + w * 2560;
    const int n = l & 31, hi = l >> 5;
    // value of (point p, feature f) = p * 32 + f
    unsigned acc = 0;
    int wa[4];
    v4s wv[4];
    for (int k = 0; k < 4; ++k) {
        const int f0 = 16 * (k >> 1) + 8 * (k & 1) + 4 * hi;
        wa[k] = piece_addr<LAYOUT>(n, f0 >> 2);
        wv[k] = v4s{(short)(n * 32 + f0), (short)(n * 32 + f0 + 1), (short)(n * 32 + f0 + 2), (short)(n * 32 + f0 + 3)};
    }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)                      // 16 stores per iteration, nothing else
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<v4s*>(tile + wa[k]) = wv[k];
        asm volatile("" ::: "memory");
    }
    for (int k = 0; k < 4; ++k) {
        v4s v = wv[k];
        for (int j = 0; j < 4; ++j) v[j] += (short)(iters - 1);
        *reinterpret_cast<v4s*>(tile + wa[k]) = v;
    }
    __syncthreads();
    long long t1 = clock64();
    const int a00 = read_addr<LAYOUT>(l, 0, 0), a01 = read_addr<LAYOUT>(l, 0, 1);
    const int a10 = read_addr<LAYOUT>(l, 1, 0), a11 = read_addr<LAYOUT>(l, 1, 1);
    v4s r00, r01, r10, r11;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                    // 16 reads per iteration; results consumed once per iteration
            r00 = tr_read(tile, a00); r01 = tr_read(tile, a01); r10 = tr_read(tile, a10); r11 = tr_read(tile, a11);
            asm volatile("" : "+v"(r00), "+v"(r01), "+v"(r10), "+v"(r11));
        }
        acc ^= (unsigned)r00[0] + (unsigned)r01[1] + (unsigned)r10[2] + (unsigned)r11[3];
    }
    long long t2 = clock64();
    if (w == 0) {
        for (int j = 0; j < 4; ++j) {
            out[l * 16 + 0 + j] = r00[j]; out[l * 16 + 4 + j] = r01[j];
            out[l * 16 + 8 + j] = r10[j]; out[l * 16 + 12 + j] = r11[j];
        }
        if (l == 0) { clocks[0] = t1 - t0; clocks[1] = t2 - t1; clocks[2] = acc; }
    }
}

template <int LAYOUT> static int run_tile(const char* name, short* d_out, long long* d_clk) {
    const int iters = 2000;
    int bad = 0;
    for (int nw : {1, 4}) {
        hipLaunchKernelGGL(tile_kernel<LAYOUT>, dim3(1), dim3(64 * nw), 0, 0, d_out, d_clk, iters);
        (void)hipDeviceSynchronize();
        short h[64 * 16]; long long c[3];
        (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(c, d_clk, sizeof c, hipMemcpyDeviceToHost);
        bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int chunk = 0; chunk < 2; ++chunk)
                for (int t = 0; t < 2; ++t)
                    for (int j = 0; j < 4; ++j) {
                        const int p = 16 * chunk + 8 * (l >> 5) + 4 * t + j, f = l & 31;
                        const short want = (short)(p * 32 + f + iters - 1);
                        bad += h[l * 16 + 8 * chunk + 4 * t + j] != want;
                    }
        printf("%-28s waves %d: operand image wrong elements %d / 1024; %6.1f clk per 4 ds_write_b64 (one tile), %6.1f clk per 4 tr reads (one operand) [16 in flight per loop trip]\n",
               name, nw, bad, (double)c[0] / iters / 4, (double)c[1] / iters / 4);
    }
    return bad;
}

int main() {
    int* d_addr; short* d_out; long long* d_clk;
    (void)hipMalloc(&d_addr, 64 * 4); (void)hipMalloc(&d_out, 64 * 16 * 2); (void)hipMalloc(&d_clk, 3 * 8);
    int rc = 0;
    for (int variant = 0; variant < 3; ++variant) {
        int addr[64];
        if (variant == 0) for (int l = 0; l < 64; ++l) addr[l] = 8 * l;
        else {                                          // a random permutation of the 8-byte pieces of an 8-KiB region
            std::vector<int> perm(1024);
            for (int i = 0; i < 1024; ++i) perm[i] = i;
            srand(17 + variant);
            for (int i = 1023; i > 0; --i) { int k = rand() % (i + 1); std::swap(perm[i], perm[k]); }
            for (int l = 0; l < 64; ++l) addr[l] = 8 * perm[l];
        }
        (void)hipMemcpy(d_addr, addr, sizeof addr, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        (void)hipDeviceSynchronize();
        short h[256];
        (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int g = l >> 4, i = l & 15;
                const int want = addr[16 * g + 4 * j + (i >> 2)] / 2 + (i & 3);
                bad += h[l * 4 + j] != (short)want;
            }
        printf("part 1, %s addresses: hypothesis H wrong for %d / 256 elements\n", variant == 0 ? "contiguous" : "permuted", bad);
        if (bad || variant == 0) {
            printf("  raw (lane: 4 element indices; supplied piece = element index / 4):\n");
            for (int l = 0; l < (bad ? 64 : 20); ++l)
                printf("  lane %2d (addr elem %4d): %5d %5d %5d %5d\n", l, addr[l] / 2, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        }
        rc |= bad != 0;
    }
    rc |= run_tile<0>("swizzled 64-B rows", d_out, d_clk) != 0;
    rc |= run_tile<1>("natural 64-B rows", d_out, d_clk) != 0;
    rc |= run_tile<2>("padded 72-B rows", d_out, d_clk) != 0;
    rc |= run_tile<3>("quad-swizzled 64-B rows", d_out, d_clk) != 0;
    printf(rc ? "FAILED\n" : "OK\n");
    return rc;
}
