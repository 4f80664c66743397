// Fused implicit decoder: plane-feature gather + all requested LocalDecoder heads in ONE kernel.
//
// Replaces, per head, reference LocalDecoder.forward (ConvONets/conv_onet/models/decoder.py:133-176):
//   3x F.grid_sample (decoder.py:117-122) -> cat -> fc_p -> 5x(fc_c + ResnetBlockFC) -> fc_out
// and the head epilogues of ConvolutionalOccupancyNetwork.decode (models/__init__.py:111-124).
// The reference re-gathers the 96-d feature for every head; here it is gathered once per point,
// kept in registers as MFMA B-operands and reused by every head.
//
// Structure (CDNA4): one wave owns T tiles of 32 query points.  Every layer is computed TRANSPOSED,
// D[feature][point] = W[feature][k] * X[k][point]: weights are the MFMA A operand (streamed from an
// LDS image of the head's fragment blob), activations the B operand.  In that orientation the D
// registers of a lane (16 features of ONE point) are, after ReLU (+ f16 rounding), exactly the B
// operand of the next layer for a permuted contraction order that is baked into the packed weights,
// so the whole 11-layer chain runs without any cross-lane traffic.  The fp32 residual stream lives
// in the accumulator (C-in = stream), biases ride in a constant-one k-slot or in the C operand.
#include "giga_dev.h"

namespace giga {

struct DecArgs {
    const void* planes;      // [3][B][40][40][32]  (plane, scene, H, W, C)  half or float
    const float* p;          // [P][3]
    const uint8_t* blob;     // packed weights
    size_t head_off[NHEADS]; // byte offset of each requested head's blob (this precision)
    int head_id[NHEADS];     // 0 qual, 1 rot, 2 width, 3 tsdf
    float* out[NHEADS];      // output pointer per requested head
    int nheads;
    int B, N;                // P = B*N points; point g belongs to scene g / N
    long long P;
    int nbatch;              // number of workgroup batches
    int post;                // 1: sigmoid(qual), normalize(rot)  (models/__init__.py:120-122)
};

__device__ __forceinline__ void store_head(const DecArgs& a, int h, long long g, float d0, float d1,
                                           float d2, float d3) {
    const int id = a.head_id[h];
    float* o = a.out[h];
    if (id == 1) {
        if (a.post) {   // F.normalize(dim=2): x / max(||x||_2, 1e-12)
            float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
            float inv = 1.0f / fmaxf(nrm, 1e-12f);
            d0 *= inv; d1 *= inv; d2 *= inv; d3 *= inv;
        }
        *reinterpret_cast<float4*>(o + 4 * g) = make_float4(d0, d1, d2, d3);
    } else {
        if (id == 0 && a.post) d0 = 1.0f / (1.0f + expf(-d0));
        o[g] = d0;
    }
}

__device__ __forceinline__ void stage_blob(uint8_t* smem, const uint8_t* src, int bytes) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

// =============================== f16 MFMA path =====================================================
template <int T>
__global__ __launch_bounds__(256, 2) void decoder_f16_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const half8* W = reinterpret_cast<const half8*>(smem);
    const float* ctab = reinterpret_cast<const float*>(smem + (size_t)DEC16_FRAGS * FRAG);
    const half_t* planes = reinterpret_cast<const half_t*>(a.planes);
    const size_t plane_stride = (size_t)a.B * RES * RES * CD;

    for (int batch = blockIdx.x; batch < a.nbatch; batch += gridDim.x) {
        // ---------------- gather: 96 features -> 6 B-operand chunks per tile ----------------------
        half8 cf[T][6], ax[T];
        long long gidx[T];
        bool valid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long g = ((long long)batch * 4 * T + wave * T + t) * 32 + n;
            valid[t] = g < a.P;
            if (!valid[t]) g = a.P - 1;
            gidx[t] = g;
            const float px = a.p[3 * g + 0], py = a.p[3 * g + 1], pz = a.p[3 * g + 2];
            const int b = (int)(g / a.N);
            const float nx = norm_coord(px), ny = norm_coord(py), nz = norm_coord(pz);
            // aux chunk: [p_hi(3), 1, p_lo(3), 1] on hi=0 lanes, [p_hi(3), 0...] on hi=1 lanes
            half_t xh = (half_t)px, yh = (half_t)py, zh = (half_t)pz;
            half8 av = {xh, yh, zh, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
            if (hi == 0) {
                av[3] = (half_t)1.0f;
                av[4] = (half_t)(px - (float)xh); av[5] = (half_t)(py - (float)yh);
                av[6] = (half_t)(pz - (float)zh); av[7] = (half_t)1.0f;
            }
            ax[t] = av;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                // xz: (u,v)=(x,z)  xy: (x,y)  yz: (y,z)      common.py:246-251
                const float u = pl == 2 ? ny : nx;
                const float v = pl == 1 ? ny : nz;
                const Bilin bl = bilin_setup(u, v);
                const half_t* base = planes + pl * plane_stride + (size_t)b * RES * RES * CD + 8 * hi;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const half8 v00 = *reinterpret_cast<const half8*>(base + (size_t)bl.o00 * CD + 16 * hf);
                    const half8 v01 = *reinterpret_cast<const half8*>(base + (size_t)bl.o01 * CD + 16 * hf);
                    const half8 v10 = *reinterpret_cast<const half8*>(base + (size_t)bl.o10 * CD + 16 * hf);
                    const half8 v11 = *reinterpret_cast<const half8*>(base + (size_t)bl.o11 * CD + 16 * hf);
                    half8 r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float acc = (float)v00[j] * bl.w00;
                        acc = fmaf((float)v01[j], bl.w01, acc);
                        acc = fmaf((float)v10[j], bl.w10, acc);
                        acc = fmaf((float)v11[j], bl.w11, acc);
                        r[j] = (half_t)acc;
                    }
                    cf[t][2 * pl + hf] = r;
                }
            }
        }
        // ---------------- heads --------------------------------------------------------------------
        for (int h = 0; h < a.nheads; ++h) {
            __syncthreads();
            stage_blob(smem, a.blob + a.head_off[h], (int)DEC16_BYTES);
            __syncthreads();
            f32x16 net[T], hh[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) net[t][r] = 0.f;
            int k = 0;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    const half8 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) net[t] = mfma16(A, c < 6 ? cf[t][c] : ax[t], net[t]);
                }
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + blk * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
                {
                    const half8 A0 = W[k * 64 + lane], A1 = W[(k + 1) * 64 + lane];
                    k += 2;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        hh[t] = mfma16(A0, pack_relu8(net[t], 0), c0);
                        hh[t] = mfma16(A1, pack_relu8(net[t], 1), hh[t]);
                    }
                }
                {
                    const half8 A0 = W[k * 64 + lane], A1 = W[(k + 1) * 64 + lane];
                    k += 2;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        net[t] = mfma16(A0, pack_relu8(hh[t], 0), net[t]);
                        net[t] = mfma16(A1, pack_relu8(hh[t], 1), net[t]);
                    }
                }
            }
            {   // + fc_1 bias of the last block, then fc_out(relu(net))
                const half8 A = W[(k++) * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) net[t] = mfma16(A, ax[t], net[t]);
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + NBLK * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
                const half8 A0 = W[k * 64 + lane], A1 = W[(k + 1) * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    f32x16 o = mfma16(A0, pack_relu8(net[t], 0), c0);
                    o = mfma16(A1, pack_relu8(net[t], 1), o);
                    if (hi == 0 && valid[t]) store_head(a, h, gidx[t], o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}

// =============================== exact fp32 MFMA path ==============================================
// Same chain on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain).  B operand of MFMA s of a hidden
// layer is simply relu(D[s]) of the previous layer: no conversion, no data movement.
template <int T>
__global__ __launch_bounds__(256, 1) void decoder_f32_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const float4* W = reinterpret_cast<const float4*>(smem);
    const float* ctab = reinterpret_cast<const float*>(smem + (size_t)DEC32_FRAGS * FRAG);
    const float* planes = reinterpret_cast<const float*>(a.planes);
    const size_t plane_stride = (size_t)a.B * RES * RES * CD;

    for (int batch = blockIdx.x; batch < a.nbatch; batch += gridDim.x) {
        float cf[T][48], ax0[T], ax1[T];
        long long gidx[T];
        bool valid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            long long g = ((long long)batch * 4 * T + wave * T + t) * 32 + n;
            valid[t] = g < a.P;
            if (!valid[t]) g = a.P - 1;
            gidx[t] = g;
            const float px = a.p[3 * g + 0], py = a.p[3 * g + 1], pz = a.p[3 * g + 2];
            const int b = (int)(g / a.N);
            const float nx = norm_coord(px), ny = norm_coord(py), nz = norm_coord(pz);
            ax0[t] = hi ? py : px;       // aux MFMA 0: slots (px, py)
            ax1[t] = hi ? 1.0f : pz;     // aux MFMA 1: slots (pz, 1)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const float u = pl == 2 ? ny : nx;
                const float v = pl == 1 ? ny : nz;
                const Bilin bl = bilin_setup(u, v);
                const float* base = planes + pl * plane_stride + (size_t)b * RES * RES * CD + 16 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v00 = *reinterpret_cast<const float4*>(base + (size_t)bl.o00 * CD + 4 * q);
                    const float4 v01 = *reinterpret_cast<const float4*>(base + (size_t)bl.o01 * CD + 4 * q);
                    const float4 v10 = *reinterpret_cast<const float4*>(base + (size_t)bl.o10 * CD + 4 * q);
                    const float4 v11 = *reinterpret_cast<const float4*>(base + (size_t)bl.o11 * CD + 4 * q);
                    // same order as aten's grid_sampler: nw, ne, sw, se
                    cf[t][16 * pl + 4 * q + 0] = fmaf(v11.x, bl.w11, fmaf(v10.x, bl.w10, fmaf(v01.x, bl.w01, v00.x * bl.w00)));
                    cf[t][16 * pl + 4 * q + 1] = fmaf(v11.y, bl.w11, fmaf(v10.y, bl.w10, fmaf(v01.y, bl.w01, v00.y * bl.w00)));
                    cf[t][16 * pl + 4 * q + 2] = fmaf(v11.z, bl.w11, fmaf(v10.z, bl.w10, fmaf(v01.z, bl.w01, v00.z * bl.w00)));
                    cf[t][16 * pl + 4 * q + 3] = fmaf(v11.w, bl.w11, fmaf(v10.w, bl.w10, fmaf(v01.w, bl.w01, v00.w * bl.w00)));
                }
            }
        }
        for (int h = 0; h < a.nheads; ++h) {
            __syncthreads();
            stage_blob(smem, a.blob + a.head_off[h], (int)DEC32_BYTES);
            __syncthreads();
            f32x16 net[T], hh[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) net[t][r] = 0.f;
            int k = 0;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        net[t] = mfma32(A.x, cf[t][4 * q + 0], net[t]);
                        net[t] = mfma32(A.y, cf[t][4 * q + 1], net[t]);
                        net[t] = mfma32(A.z, cf[t][4 * q + 2], net[t]);
                        net[t] = mfma32(A.w, cf[t][4 * q + 3], net[t]);
                    }
                }
                {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        net[t] = mfma32(A.x, ax0[t], net[t]);
                        net[t] = mfma32(A.y, ax1[t], net[t]);
                    }
                }
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + blk * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) hh[t] = c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        hh[t] = mfma32(A.x, relu(net[t][4 * q + 0]), hh[t]);
                        hh[t] = mfma32(A.y, relu(net[t][4 * q + 1]), hh[t]);
                        hh[t] = mfma32(A.z, relu(net[t][4 * q + 2]), hh[t]);
                        hh[t] = mfma32(A.w, relu(net[t][4 * q + 3]), hh[t]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        // B operands must be the PRE-update hidden values: hh is not modified here
                        net[t] = mfma32(A.x, relu(hh[t][4 * q + 0]), net[t]);
                        net[t] = mfma32(A.y, relu(hh[t][4 * q + 1]), net[t]);
                        net[t] = mfma32(A.z, relu(hh[t][4 * q + 2]), net[t]);
                        net[t] = mfma32(A.w, relu(hh[t][4 * q + 3]), net[t]);
                    }
                }
            }
            {
                const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) net[t] = mfma32(A.y, ax1[t], net[t]);   // + b1 of block 4 (slot "1.0")
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + NBLK * 32 + 8 * q + 4 * hi);
                    c0[4 * q + 0] = v.x; c0[4 * q + 1] = v.y; c0[4 * q + 2] = v.z; c0[4 * q + 3] = v.w;
                }
                f32x16 o[T];
#pragma unroll
                for (int t = 0; t < T; ++t) o[t] = c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        o[t] = mfma32(A.x, relu(net[t][4 * q + 0]), o[t]);
                        o[t] = mfma32(A.y, relu(net[t][4 * q + 1]), o[t]);
                        o[t] = mfma32(A.z, relu(net[t][4 * q + 2]), o[t]);
                        o[t] = mfma32(A.w, relu(net[t][4 * q + 3]), o[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (hi == 0 && valid[t]) store_head(a, h, gidx[t], o[t][0], o[t][1], o[t][2], o[t][3]);
            }
        }
    }
}

// ------------------------------- plane repack NCHW fp32 -> NHWC T ---------------------------------
// Used when the planes arrive through the Python boundary as the reference's (B,32,40,40) tensors
// (LocalDecoder.forward(p, c_plane), decoder.py:133).  One thread per (image, pixel, 4 channels).
template <typename TOut>
__global__ void planes_nchw_to_nhwc_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                           const float* __restrict__ src2, TOut* __restrict__ dst, int B) {
    __shared__ float tile[32][RES + 1];
    // block = one (plane, scene, row y): transposes a 32(c) x 40(x) slab
    const int img = blockIdx.x / RES, y = blockIdx.x % RES;
    const int pl = img / B, b = img % B;
    const float* src = (pl == 0 ? src0 : pl == 1 ? src1 : src2) + (size_t)b * CD * RES * RES;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int c = i / RES, x = i % RES;
        tile[c][x] = src[(size_t)c * RES * RES + y * RES + x];
    }
    __syncthreads();
    TOut* d = dst + ((size_t)img * RES * RES + (size_t)y * RES) * CD;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int x = i / CD, c = i % CD;
        d[(size_t)x * CD + c] = (TOut)tile[c][x];
    }
}

template <typename TIn>
__global__ void planes_nhwc_to_nchw_kernel(const TIn* __restrict__ src, float* __restrict__ dst) {
    __shared__ float tile[32][RES + 1];
    const int img = blockIdx.x / RES, y = blockIdx.x % RES;
    const TIn* s = src + ((size_t)img * RES * RES + (size_t)y * RES) * CD;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int x = i / CD, c = i % CD;
        tile[c][x] = (float)s[(size_t)x * CD + c];
    }
    __syncthreads();
    float* d = dst + (size_t)img * CD * RES * RES;
    for (int i = threadIdx.x; i < CD * RES; i += blockDim.x) {
        const int c = i / RES, x = i % RES;
        d[(size_t)c * RES * RES + y * RES + x] = tile[c][x];
    }
}

// ------------------------------- launchers ----------------------------------------------------------
int launch_decoder(const DecArgs& a0, int precision, hipStream_t s, void* ev0, void* ev1) {
    DecArgs a = a0;
    if (a.P <= 0 || a.nheads <= 0) return 0;
    if (ev0 && ev1) (void)hipEventRecord(static_cast<hipEvent_t>(ev0), s);
    const long long tiles = (a.P + 31) / 32;
    if (precision == 1) {
        constexpr int T = 2;
        a.nbatch = (int)((tiles + 4 * T - 1) / (4 * T));
        const int grid = a.nbatch < 512 ? a.nbatch : 512;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_f16_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC16_BYTES);
        hipLaunchKernelGGL(decoder_f16_kernel<T>, dim3(grid), dim3(256), DEC16_BYTES, s, a);
    } else {
        constexpr int T = 1;
        a.nbatch = (int)((tiles + 4 * T - 1) / (4 * T));
        const int grid = a.nbatch < 256 ? a.nbatch : 256;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_f32_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC32_BYTES);
        hipLaunchKernelGGL(decoder_f32_kernel<T>, dim3(grid), dim3(256), DEC32_BYTES, s, a);
    }
    if (ev0 && ev1) (void)hipEventRecord(static_cast<hipEvent_t>(ev1), s);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_planes_pack(const float* xz, const float* xy, const float* yz, void* dst, int B, int precision,
                       hipStream_t s) {
    if (B <= 0) return 0;
    if (precision == 1)
        hipLaunchKernelGGL(planes_nchw_to_nhwc_kernel<half_t>, dim3(3 * B * RES), dim3(256), 0, s, xz, xy, yz,
                           reinterpret_cast<half_t*>(dst), B);
    else
        hipLaunchKernelGGL(planes_nchw_to_nhwc_kernel<float>, dim3(3 * B * RES), dim3(256), 0, s, xz, xy, yz,
                           reinterpret_cast<float*>(dst), B);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_planes_unpack(const void* src, float* dst, int B, int precision, hipStream_t s) {
    if (B <= 0) return 0;
    if (precision == 1)
        hipLaunchKernelGGL(planes_nhwc_to_nchw_kernel<half_t>, dim3(3 * B * RES), dim3(256), 0, s,
                           reinterpret_cast<const half_t*>(src), dst);
    else
        hipLaunchKernelGGL(planes_nhwc_to_nchw_kernel<float>, dim3(3 * B * RES), dim3(256), 0, s,
                           reinterpret_cast<const float*>(src), dst);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
