// Decoder backward (training path, fp32): gradients of every LocalDecoder parameter and of the feature
// planes (reference: autograd through conv_onet/models/decoder.py:117-176, layers.py:39-47 and the epilogues
// of models/__init__.py:111-124, driven by scripts/train_giga.py:198-211).
//
// decoder_bwd_kernel, per 32-point tile and head:
//   1. recompute the forward chain (same fragments / MFMAs as decoder_f32_kernel), keeping the stream before
//      every block (net_i') and every hidden activation (h_i) in registers;
//   2. swap the LDS image to the TRANSPOSED matrices (backward blob) and run the gradient chain
//         G = DN[i+1];  DH[i] = (W1_i^T G) * (h_i > 0);  DN[i] = G + (W0_i^T DH[i]) * (net_i' > 0);  dc += Wc_i^T DN[i]
//      in the same transposed layout (lane = point, registers = features), so again no cross-lane traffic;
//   3. write the (dY, X) pairs of every linear layer to HBM as [point][32] rows and scatter dc into the
//      plane gradients with the bilinear weights (fp32 atomics, like aten's grid_sampler backward; transposed
//      through LDS so that an atomic instruction touches 128-byte runs of channels).
// linear_wgrad_kernel then reduces dW = dY^T X and db = colsum(dY) for all layers of a head in ONE launch
// (MFMA 32x32x2 over pairs of points, operands straight from HBM, one atomic per gradient element per block).
#include "giga_args.h"
#include "giga_side.h"
#include "giga_dev.h"

namespace giga {

// per-head scratch arrays, each [P][32] fp32 (row = point, D-layout feature order restored to 0..31)
//   0..5 DN[i]   6..10 DH[i]   11..15 XN[i]   16..20 XH[i]   21 XO   22 DO (cols >= out_dim zero)
constexpr int DB_NARR = 23;
// shared per call: C [P][96], PP [P][32] (cols 0..2 = p, rest zero)

struct DecBwdArgs {
    const float* planes;        // NHWC fp32 [3][B][40][40][32]
    const float* p;             // [P][3]
    const uint8_t* blob;        // forward blob (fp32 images)
    const uint8_t* bwd_blob;    // backward blob (transposed matrices)
    size_t head_off[NHEADS], bwd_off[NHEADS];
    int head_id[NHEADS];
    const float* out[NHEADS];   // forward outputs of the head (post-epilogue), needed for sigmoid / normalize backward
    const float* dout[NHEADS];  // upstream gradients, same shapes
    float* scratch[NHEADS];     // DB_NARR arrays of [P][32]
    float* Cbuf; float* Pbuf;   // [P][96], [P][32]
    float* gplanes;             // NHWC fp32 [3][B][40][40][32], accumulated with atomics; nullptr = detached head
    float* dcbuf;               // [P][96] or nullptr.  Set (many points per scene, all heads in one workgroup): the kernel stores
                                // the gradient of the sampled features per point here instead of scattering it with atomics, and
                                // plane_gather_kernel builds the plane gradients from it (below)
    int nheads, B, N;
    long long P;
    int nbatch; float invN;
    int heads_per_wg;           // heads handled by one workgroup (blockIdx.y selects the group): 1 when the point batches cannot fill the chip
};

template <int CHUNKS>
__device__ __forceinline__ void dma_image4(const uint8_t* src, uint8_t* lds_dst, int wave, int lane) {
    for (int c = wave; c < CHUNKS; c += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * FRAG + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds_dst + c * FRAG), 16, 0, 0);
}

// write a D-layout register image (lane = point n, registers = features drow(r,hi)) as row `g` of a [P][32] array
__device__ __forceinline__ void store_rows(float* arr, long long g, int hi, const f32x16& v, bool ok) {
    if (!ok) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(arr + g * 32 + 8 * q + 4 * hi) =
            make_float4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

__global__ __launch_bounds__(256, 1) void decoder_bwd_kernel(DecBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const float4* W = reinterpret_cast<const float4*>(smem);
    const size_t plane_stride = (size_t)a.B * RES * RES * CD;

    for (int bseq = blockIdx.x; bseq < a.nbatch; bseq += gridDim.x) {
        const int batch = xcd_swizzle(bseq, a.nbatch);        // an XCD works on a contiguous range of points (L2 locality)
        // ---------------- gather (as decoder_f32_kernel), keep the bilinear footprints for the scatter --------
        long long g = ((long long)batch * 4 + wave) * 32 + n;
        const bool valid = g < a.P;
        if (!valid) g = a.P - 1;
        int b, rdummy;
        split_scene(g, a.N, a.invN, b, rdummy);
        const float px = a.p[3 * g + 0], py = a.p[3 * g + 1], pz = a.p[3 * g + 2];
        const float nx = norm_coord(px), ny = norm_coord(py), nz = norm_coord(pz);
        const float ax0 = hi ? py : px, ax1 = hi ? 1.0f : pz, ax2 = hi ? 0.0f : 1.0f;
        Bilin bl[3];
        bl[0] = bilin_setup(nx, nz); bl[1] = bilin_setup(nx, ny); bl[2] = bilin_setup(ny, nz);
        float cf[48];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const float* base = a.planes + pl * plane_stride + (size_t)b * RES * RES * CD + 16 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v00 = *reinterpret_cast<const float4*>(base + (size_t)bl[pl].o00 * CD + 4 * q);
                const float4 v01 = *reinterpret_cast<const float4*>(base + (size_t)bl[pl].o01 * CD + 4 * q);
                const float4 v10 = *reinterpret_cast<const float4*>(base + (size_t)bl[pl].o10 * CD + 4 * q);
                const float4 v11 = *reinterpret_cast<const float4*>(base + (size_t)bl[pl].o11 * CD + 4 * q);
                cf[16 * pl + 4 * q + 0] = fmaf(v11.x, bl[pl].w11, fmaf(v10.x, bl[pl].w10, fmaf(v01.x, bl[pl].w01, v00.x * bl[pl].w00)));
                cf[16 * pl + 4 * q + 1] = fmaf(v11.y, bl[pl].w11, fmaf(v10.y, bl[pl].w10, fmaf(v01.y, bl[pl].w01, v00.y * bl[pl].w00)));
                cf[16 * pl + 4 * q + 2] = fmaf(v11.z, bl[pl].w11, fmaf(v10.z, bl[pl].w10, fmaf(v01.z, bl[pl].w01, v00.z * bl[pl].w00)));
                cf[16 * pl + 4 * q + 3] = fmaf(v11.w, bl[pl].w11, fmaf(v10.w, bl[pl].w10, fmaf(v01.w, bl[pl].w01, v00.w * bl[pl].w00)));
            }
        }
        // X of fc_c / fc_p: rows of C [P][96] (feature pl*32 + 16*hi + s) and PP [P][32]
        if (valid) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.Cbuf + g * 96 + pl * 32 + 16 * hi + 4 * q) =
                        make_float4(cf[16 * pl + 4 * q], cf[16 * pl + 4 * q + 1], cf[16 * pl + 4 * q + 2], cf[16 * pl + 4 * q + 3]);
            if (hi == 0) {
                float* pr = a.Pbuf + g * 32;
                *reinterpret_cast<float4*>(pr) = make_float4(px, py, pz, 0.f);
#pragma unroll
                for (int q = 1; q < 8; ++q) *reinterpret_cast<float4*>(pr + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        f32x16 dc[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int r = 0; r < 16; ++r) dc[pl][r] = 0.f;

        // few point batches (train_giga's single grasp query per scene = ONE batch): every head gets its own workgroup -- the
        // three grasp heads of 32 points took 120 us one after the other in a single workgroup, 40 us side by side; the plane
        // gradient is accumulated with atomics either way
        const int h_begin = (int)blockIdx.y * a.heads_per_wg;
        const int h_end = h_begin + a.heads_per_wg < a.nheads ? h_begin + a.heads_per_wg : a.nheads;
        for (int h = h_begin; h < h_end; ++h) {
            float* S = a.scratch[h];
            const size_t AS = (size_t)a.P * 32;                     // array stride
            // ================= 1. forward recompute (fragment order of giga_pack.cpp::pack_head32) ==========
            __syncthreads();
            dma_image4<(int)(DEC32_BYTES / FRAG)>(a.blob + a.head_off[h], smem, wave, lane);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            const float* ctab = reinterpret_cast<const float*>(smem + (size_t)DEC32_FRAGS * FRAG);
            // the backward chain needs only the SIGNS of the pre-activations: 16 bits per lane and tensor instead of 16
            // registers; the activations themselves (X of the weight-gradient GEMMs) leave as soon as they exist
            unsigned mnet[NBLK + 1], mhid[NBLK];
            f32x16 net;
#pragma unroll
            for (int r = 0; r < 16; ++r) net[r] = 0.f;
            int k = 0;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
                    net = mfma32(A.x, cf[4 * q + 0], net); net = mfma32(A.y, cf[4 * q + 1], net);
                    net = mfma32(A.z, cf[4 * q + 2], net); net = mfma32(A.w, cf[4 * q + 3], net);
                }
                {
                    const float4 A = W[(k++) * 64 + lane];
                    net = mfma32(A.x, ax0, net); net = mfma32(A.y, ax1, net); net = mfma32(A.z, ax2, net);
                }
                {
                    f32x16 t;
                    unsigned m = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { t[r] = relu_ieee(net[r]); m |= (net[r] > 0.f ? 1u : 0u) << r; }
                    mnet[blk] = m;
                    store_rows(S + (11 + blk) * AS, g, hi, t, valid);          // XN[blk]
                }
                f32x16 hcur;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(ctab + blk * 32 + 8 * q + 4 * hi);
                    hcur[4 * q + 0] = v.x; hcur[4 * q + 1] = v.y; hcur[4 * q + 2] = v.z; hcur[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
                    hcur = mfma32(A.x, relu_ieee(net[4 * q + 0]), hcur); hcur = mfma32(A.y, relu_ieee(net[4 * q + 1]), hcur);
                    hcur = mfma32(A.z, relu_ieee(net[4 * q + 2]), hcur); hcur = mfma32(A.w, relu_ieee(net[4 * q + 3]), hcur);
                }
                {
                    f32x16 t;
                    unsigned m = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { t[r] = relu_ieee(hcur[r]); m |= (hcur[r] > 0.f ? 1u : 0u) << r; }
                    mhid[blk] = m;
                    store_rows(S + (16 + blk) * AS, g, hi, t, valid);          // XH[blk]
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 A = W[(k++) * 64 + lane];
                    net = mfma32(A.x, relu_ieee(hcur[4 * q + 0]), net); net = mfma32(A.y, relu_ieee(hcur[4 * q + 1]), net);
                    net = mfma32(A.z, relu_ieee(hcur[4 * q + 2]), net); net = mfma32(A.w, relu_ieee(hcur[4 * q + 3]), net);
                }
            }
            {   // + fc_1 bias of the last block -> net5
                const float4 A = W[(k++) * 64 + lane];
                net = mfma32(A.y, ax1, net);
            }
            {   // XO = relu(net5) and its sign mask
                f32x16 t;
                unsigned m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) { t[r] = relu_ieee(net[r]); m |= (net[r] > 0.f ? 1u : 0u) << r; }
                mnet[NBLK] = m;
                store_rows(S + 21 * AS, g, hi, t, valid);
            }
            // ================= 2. epilogue backward: dO (<= 4 values, identical in both lane halves) ============
            const int id = a.head_id[h];
            float dO[4] = {0.f, 0.f, 0.f, 0.f};
            if (id == 1) {          // rot = z / max(|z|, eps): dz = (dr - r (r.dr)) / |z|   (F.normalize backward)
                const float4 r4 = *reinterpret_cast<const float4*>(a.out[h] + 4 * g);
                const float4 d4 = *reinterpret_cast<const float4*>(a.dout[h] + 4 * g);
                // |z| from the recomputed raw output would need fc_out; use r = z/|z| and z.r = |z|: recompute z below
                dO[0] = d4.x; dO[1] = d4.y; dO[2] = d4.z; dO[3] = d4.w;
                (void)r4;
            } else {
                float d = a.dout[h][g];
                if (id == 0) { const float q = a.out[h][g]; d *= q * (1.0f - q); }      // sigmoid'
                dO[0] = d;
            }
            // ================= 3. swap to the transposed image ====================================================
            __syncthreads();
            dma_image4<(int)(DECB_BYTES / FRAG)>(a.bwd_blob + a.bwd_off[h], smem, wave, lane);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            const float* wout = reinterpret_cast<const float*>(smem + (size_t)DECB_FRAGS * FRAG);   // [4][32]
            if (id == 1) {
                // raw z = fc_out(relu(net5)) + b is needed for the normalize backward: z_o = sum_f Wout[o][f] relu(net5)[f] + b_o
                // (this lane holds 16 of the 32 features; the other half lives in lane^32)
                float z[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s = fmaf(wout[o * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi], relu_ieee(net[r]), s);
                    s += __shfl_xor(s, 32);
                    z[o] = s;
                }
                // bias of fc_out: forward C-table row 5 is gone from LDS; the caller passes post-normalize outputs,
                // and r = z/|z| with |z| = z.r, so recover |z| from the bias-free part plus the known direction:
                // z = zb + bias  =>  we instead read the bias from the forward blob in global memory
                const float* fb = reinterpret_cast<const float*>(a.blob + a.head_off[h] + (size_t)DEC32_FRAGS * FRAG) + NBLK * 32;
                const float4 r4 = *reinterpret_cast<const float4*>(a.out[h] + 4 * g);
                float nrm2 = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o) { z[o] += fb[o]; nrm2 = fmaf(z[o], z[o], nrm2); }
                const float inv = 1.0f / fmaxf(sqrtf(nrm2), 1e-12f);
                const float rd = r4.x * dO[0] + r4.y * dO[1] + r4.z * dO[2] + r4.w * dO[3];
                dO[0] = (dO[0] - r4.x * rd) * inv; dO[1] = (dO[1] - r4.y * rd) * inv;
                dO[2] = (dO[2] - r4.z * rd) * inv; dO[3] = (dO[3] - r4.w * rd) * inv;
            }
            if (valid && hi == 0) {       // DO row: [dO0..3, 0...]
                float* dr = S + 22 * AS + g * 32;
                *reinterpret_cast<float4*>(dr) = make_float4(dO[0], dO[1], dO[2], dO[3]);
#pragma unroll
                for (int q = 1; q < 8; ++q) *reinterpret_cast<float4*>(dr + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // DN[5] = (Wout^T dO) * (net5 > 0)
            f32x16 G;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float v = wout[f] * dO[0] + wout[32 + f] * dO[1] + wout[64 + f] * dO[2] + wout[96 + f] * dO[3];
                G[r] = (mnet[NBLK] >> r & 1u) ? v : 0.f;
            }
            store_rows(S + 5 * AS, g, hi, G, valid);
            // ================= 4. gradient chain (fragments: block b -> Wc^T 20b..20b+11, W0^T +12..15, W1^T +16..19) ====
#pragma unroll
            for (int blk = NBLK - 1; blk >= 0; --blk) {
                const int kb = 20 * blk;
                f32x16 dh;
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                       // W1^T G
                    const float4 A = W[(kb + 16 + q) * 64 + lane];
                    dh = mfma32(A.x, G[4 * q + 0], dh); dh = mfma32(A.y, G[4 * q + 1], dh);
                    dh = mfma32(A.z, G[4 * q + 2], dh); dh = mfma32(A.w, G[4 * q + 3], dh);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[r] = (mhid[blk] >> r & 1u) ? dh[r] : 0.f;
                store_rows(S + (6 + blk) * AS, g, hi, dh, valid);
                f32x16 dn;
#pragma unroll
                for (int r = 0; r < 16; ++r) dn[r] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                       // W0^T DH
                    const float4 A = W[(kb + 12 + q) * 64 + lane];
                    dn = mfma32(A.x, dh[4 * q + 0], dn); dn = mfma32(A.y, dh[4 * q + 1], dn);
                    dn = mfma32(A.z, dh[4 * q + 2], dn); dn = mfma32(A.w, dh[4 * q + 3], dn);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) G[r] += (mnet[blk] >> r & 1u) ? dn[r] : 0.f;     // DN[blk]
                store_rows(S + blk * AS, g, hi, G, valid);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)                      // dc += Wc^T DN[blk]
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 A = W[(kb + 4 * pl + q) * 64 + lane];
                        dc[pl] = mfma32(A.x, G[4 * q + 0], dc[pl]); dc[pl] = mfma32(A.y, G[4 * q + 1], dc[pl]);
                        dc[pl] = mfma32(A.z, G[4 * q + 2], dc[pl]); dc[pl] = mfma32(A.w, G[4 * q + 3], dc[pl]);
                    }
            }
        }
        // ---------------- scatter dc into the plane gradients (sample_plane_feature backward) ------------------
        // Every atomic instruction covers two (point, tap) pairs x 32 CONTIGUOUS channels (two 128-B runs) instead of
        // 64 scattered words: float atomics are resolved outside the XCD-local L2, one fabric operation per touched
        // line, so the transposition through LDS (the weight image is dead by now) cuts that traffic 16-fold.
        if (a.gplanes && a.dcbuf) {               // the scatter is plane_gather_kernel's: hand over dc as rows of [P][96]
            if (valid) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(a.dcbuf + g * 96 + pl * 32 + 8 * q + 4 * hi) =
                            make_float4(dc[pl][4 * q], dc[pl][4 * q + 1], dc[pl][4 * q + 2], dc[pl][4 * q + 3]);
            }
        } else if (a.gplanes) {                   // nullptr: this head is detached from the planes
            __syncthreads();                      // every wave is done with the weight image
            float* T = reinterpret_cast<float*>(smem) + wave * (32 * 96 + 32 * 12 * 2);    // [point][96] values
            int* Q = reinterpret_cast<int*>(T + 32 * 96);                                   // [point][plane][tap] offsets
            float* Wt = reinterpret_cast<float*>(Q + 32 * 12);                              // [point][plane][tap] weights
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(T + n * 96 + pl * 32 + 8 * q + 4 * hi) =
                        make_float4(dc[pl][4 * q], dc[pl][4 * q + 1], dc[pl][4 * q + 2], dc[pl][4 * q + 3]);
                if (hi == 0) {
                    const int base = valid ? (int)(pl * plane_stride + (size_t)b * RES * RES * CD) : -1;
                    const int o4[4] = {bl[pl].o00, bl[pl].o01, bl[pl].o10, bl[pl].o11};
                    const float w4[4] = {bl[pl].w00, bl[pl].w01, bl[pl].w10, bl[pl].w11};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        Q[(n * 3 + pl) * 4 + t] = valid ? base + o4[t] * CD : -1;
                        Wt[(n * 3 + pl) * 4 + t] = w4[t];
                    }
                }
            }
            // wave-private staging: DS operations of one wave execute in order, no barrier needed
            const int c = lane & 31;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                for (int pr = 0; pr < 16; ++pr) {
                    const int pt = 2 * pr + hi;
                    const float v = T[pt * 96 + pl * 32 + c];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int off = Q[(pt * 3 + pl) * 4 + t];
                        const float w = Wt[(pt * 3 + pl) * 4 + t];
                        if (off >= 0) atomicAdd(a.gplanes + off + c, v * w);
                    }
                }
        }
    }
}

// ------------------------------- plane gradients of a many-point head, without float atomics --------------------
// sample_plane_feature backward (decoder.py:117-122 through autograd = aten's grid_sampler_2d backward): every query point adds
// w_tap * dc[plane][0..31] to the four pixels of its bilinear footprint in each of the three planes.  As fp32 atomics on HBM
// that was 74 of the 170 us of the occupancy head's backward at 32 x 2048 queries (25 M atomic lanes; float atomics are resolved
// outside the XCD-local L2).  LDS float atomics are no way out either: ds_add_f32 retires about one LANE per clock (a
// 40 x 40 x 16 tile per workgroup filled that way took 169 us, profiles/r03/).  So the scatter is turned into a GATHER:
// a workgroup owns (half of the pixels, plane, scene); it bins the scene's N points by the pixel cell of their footprint's
// corner (counting sort in LDS: integer atomics for the histogram and the ranks, one wave for the prefix sum, every cell's short
// list then ordered by point index), and every output pixel sums the contributions of the points in its four neighbouring cells,
// with the weights bilin_setup gives (the forward's own).  The result is WRITTEN (no memset, no read-modify-write): a head that
// uses this path runs before the heads that add with atomics.  Unlike atomics, the sum is in a fixed order (by point index
// inside a cell, cells in a fixed order): the plane gradient of this head is reproducible bit for bit.
constexpr int PS_NW = 16;                                   // waves per workgroup
constexpr int PS_MAXN = 4096;                               // points per scene this path takes (LDS: 6 + 16 bytes per point)
constexpr int PS_CELLS = RES * RES;
constexpr size_t PS_LDS = (size_t)(2 * PS_CELLS + 64) * 4 + (size_t)PS_MAXN * (6 + 16);
__global__ __launch_bounds__(PS_NW * 64) void plane_gather_kernel(const float* __restrict__ dcbuf, const float* __restrict__ p,
                                                                  float* __restrict__ gplanes, int B, int N) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float4* wts = reinterpret_cast<float4*>(smem);                    // [N] (w00, w01, w10, w11) of every point, by point index
    int* cnt = reinterpret_cast<int*>(wts + PS_MAXN);                 // [1600] histogram
    int* start = cnt + PS_CELLS;                                      // [1601] first slot of every cell
    unsigned short* cell_of = reinterpret_cast<unsigned short*>(start + PS_CELLS + 64);    // [N]
    unsigned short* rank_of = cell_of + PS_MAXN;                      // [N]
    unsigned short* sorted = rank_of + PS_MAXN;                       // [N] point indices, cell by cell
    const int half = blockIdx.x, pl = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t g0 = (size_t)b * N;
    for (int i = tid; i < PS_CELLS; i += PS_NW * 64) cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += PS_NW * 64) {
        const size_t g = g0 + i;
        const float nx = norm_coord(p[3 * g + 0]), ny = norm_coord(p[3 * g + 1]), nz = norm_coord(p[3 * g + 2]);
        const Bilin bl = pl == 0 ? bilin_setup(nx, nz) : pl == 1 ? bilin_setup(nx, ny) : bilin_setup(ny, nz);
        wts[i] = make_float4(bl.w00, bl.w01, bl.w10, bl.w11);
        cell_of[i] = (unsigned short)bl.o00;                          // y0 * 40 + x0
        rank_of[i] = (unsigned short)atomicAdd(cnt + bl.o00, 1);
    }
    __syncthreads();
    if (wave == 0) {                                                  // exclusive prefix sum of the 1600 counts: 25 cells per lane
        int tot = 0;
        for (int k = 0; k < 25; ++k) tot += cnt[lane * 25 + k];
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        int run = incl - tot;
        for (int k = 0; k < 25; ++k) { start[lane * 25 + k] = run; run += cnt[lane * 25 + k]; }
        if (lane == 63) start[PS_CELLS] = run;
    }
    __syncthreads();
    for (int i = tid; i < N; i += PS_NW * 64) sorted[start[cell_of[i]] + rank_of[i]] = (unsigned short)i;
    __syncthreads();
    // order every cell's list by point index.  A short list (the normal case: 2.6 points per cell at 2048 queries) is sorted by
    // insertion by the thread that owns the cell; LONG lists -- queries clustered in one cell: points clamped onto a face of the
    // cube, duplicates -- would be O(n^2) serial LDS work on one lane (8 M steps at 4096 points), so they are ranked by the whole
    // workgroup instead: element x goes to slot #{y : list[y] < list[x]} (point indices are distinct), n^2 / 1024 steps per thread.
    constexpr int PS_LONG = 32;
    int* nlong = start + PS_CELLS + 1;                                // (the 63 spare words behind start[]): [0] count, [1..] long cells
    if (tid == 0) nlong[0] = 0;
    __syncthreads();
    for (int c = tid; c < PS_CELLS; c += PS_NW * 64) {
        const int s0 = start[c], s1 = start[c + 1];
        if (s1 - s0 > PS_LONG) {                                      // (at most 4096 / 33 = 124 such cells; the first 62 are listed,
            const int k = atomicAdd(nlong, 1);                        //  the others fall back to the serial sort below)
            if (k < 62) { nlong[1 + k] = c; continue; }
        }
        for (int x = s0 + 1; x < s1; ++x) {
            const unsigned short v = sorted[x];
            int y = x - 1;
            while (y >= s0 && sorted[y] > v) { sorted[y + 1] = sorted[y]; --y; }
            sorted[y + 1] = v;
        }
    }
    __syncthreads();
    {
        const int nl = nlong[0] < 62 ? nlong[0] : 62;
        unsigned short* tmp = rank_of;                                // (the ranks were consumed by the scatter above)
        for (int k = 0; k < nl; ++k) {
            const int c = nlong[1 + k], s0 = start[c], s1 = start[c + 1];
            for (int x = s0 + tid; x < s1; x += PS_NW * 64) {
                const unsigned short v = sorted[x];
                int r = 0;
                for (int y = s0; y < s1; ++y) r += sorted[y] < v;     // (all lanes read the same word: a broadcast)
                tmp[s0 + r] = v;
            }
        }
        __syncthreads();
        for (int k = 0; k < nl; ++k) {
            const int c = nlong[1 + k], s0 = start[c], s1 = start[c + 1];
            for (int x = s0 + tid; x < s1; x += PS_NW * 64) sorted[x] = tmp[x];
        }
    }
    __syncthreads();
    // gather: 8 lanes per pixel (a float4 of channels each), this half's 800 pixels.  The cells (cy, x-1) and (cy, x) are
    // neighbours in the sorted array: one contiguous range per cell row (2.6 points on average at 2048 queries per scene); both
    // rows are walked four points at a time with all eight loads in flight.
    float* dst = gplanes + ((size_t)pl * B + b) * PS_CELLS * CD;
    const int q = tid & 7;
    for (int px = half * (PS_CELLS / 2) + (tid >> 3); px < (half + 1) * (PS_CELLS / 2); px += PS_NW * 64 / 8) {
        const int y = px / RES, x = px - y * RES;
        int k0[2], kmid[2], kend[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {                                 // r = 0: the cell row above (its taps dy = 1), r = 1: this row (dy = 0)
            const int cy = y - 1 + r;
            const int c_hi = (cy < 0 ? 0 : cy) * RES + x, c_lo = x > 0 ? c_hi - 1 : c_hi;   // tap dx = 1 comes from cell x-1
            k0[r] = start[c_lo]; kmid[r] = start[c_hi]; kend[r] = cy < 0 ? k0[r] : start[c_hi + 1];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        while (k0[0] < kend[0] || k0[1] < kend[1]) {
            float4 v[2][4], w4[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0[r] + u;
                    const int id = sorted[k < kend[r] ? k : 0];
                    w4[r][u] = wts[id];
                    v[r][u] = *reinterpret_cast<const float4*>(dcbuf + (g0 + id) * 96 + pl * 32 + 4 * q);
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0[r] + u;
                    if (k >= kend[r]) continue;
                    const bool dx = k < kmid[r];                      // from cell x-1: this pixel is its right-hand tap
                    const float w = r == 0 ? (dx ? w4[r][u].w : w4[r][u].z) : (dx ? w4[r][u].y : w4[r][u].x);
                    acc.x = fmaf(v[r][u].x, w, acc.x); acc.y = fmaf(v[r][u].y, w, acc.y);
                    acc.z = fmaf(v[r][u].z, w, acc.z); acc.w = fmaf(v[r][u].w, w, acc.w);
                }
            k0[0] += 4; k0[1] += 4;
        }
        *reinterpret_cast<float4*>(dst + (size_t)px * CD + 4 * q) = acc;
    }
}

// ------------------------------- batched dW = dY^T X, db = colsum(dY) ----------------------------------------
struct LinProblem {
    const float* R;      // dY  [P][32]
    const float* C;      // X   [P][ncol]
    float* dW;           // (mvalid x nvalid) row-major with row stride sM
    float* db;           // mvalid entries (may be null)
    int ncol, sM, mvalid, nvalid;
};
constexpr int LIN_MAX = 52;                          // 17 problems per head: up to three heads in one launch (the struct stays under the 4-KiB kernarg limit)
struct LinArgs { LinProblem pr[LIN_MAX]; int nprob; long long P; int pts_per_block; int ksplit; int nb_total; int nb_start[LIN_MAX + 1]; };

__global__ __launch_bounds__(256) void linear_wgrad_kernel(LinArgs a) {
    __shared__ float red[3][64][17];
    __shared__ float bred[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, hi = lane >> 5;
    const int blk = blockIdx.x % a.nb_total, ks = blockIdx.x / a.nb_total;
    int pi = 0;
    while (blk >= a.nb_start[pi + 1]) ++pi;
    const LinProblem& pr = a.pr[pi];
    const int nb = blk - a.nb_start[pi];                       // 32-column block of X
    const long long p0 = (long long)ks * a.pts_per_block;
    long long p1 = p0 + a.pts_per_block;
    if (p1 > a.P) p1 = a.P;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    for (long long base = p0 + 8 * wave; base < p1; base += 32) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long long p = base + 2 * s + hi;
            av[s] = p < p1 ? pr.R[p * 32 + i] : 0.f;
            bv[s] = p < p1 ? pr.C[p * pr.ncol + nb * 32 + i] : 0.f;
            bsum += av[s];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma32(av[s], bv[s], acc);
    }
    bsum += __shfl_xor(bsum, 32);
    if (hi == 0) bred[wave][i] = bsum;
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + red[0][lane][r] + red[1][lane][r] + red[2][lane][r];
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hi, nn = nb * 32 + i;
            if (m < pr.mvalid && nn < pr.nvalid) atomicAdd(pr.dW + (size_t)m * pr.sM + nn, v);
        }
        if (nb == 0 && pr.db && hi == 0 && i < pr.mvalid)
            atomicAdd(pr.db + i, bred[0][i] + bred[1][i] + bred[2][i] + bred[3][i]);
    }
}

// ------------------------------- launcher ------------------------------------------------------------------------
// plane gradients of a many-point head from its dc rows [B * N][96] (WRITTEN; the caller has checked dec_bwd_writes_planes(1, B, N))
int launch_plane_gather(const float* dcbuf, const float* p, float* gplanes, int B, int N, hipStream_t s) {
    giga::dyn_lds_once(reinterpret_cast<const void*>(plane_gather_kernel), (int)PS_LDS);
    GIGA_LAUNCH(plane_gather_kernel, dim3(2, 3, B), dim3(PS_NW * 64), PS_LDS, s, dcbuf, p, gplanes, B, N);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

size_t dec_bwd_scratch_floats(long long P, int nheads) { return (size_t)P * 32 * DB_NARR * nheads + (size_t)P * (96 + 32 + 96); }

// the plane gradients of this call are gathered and WRITTEN (plane_gather_kernel) instead of added with atomics
bool dec_bwd_writes_planes(int nheads, int B, int N) { return nheads == 1 && N >= 256 && N <= PS_MAXN && B > 0; }

int launch_decoder_backward(const float* planes, const float* p, const uint8_t* blob, const uint8_t* bwd_blob,
                            int head_mask, const float* const* outs, const float* const* douts, float* gplanes,
                            float* grads, int head_present, float* scratch, int B, int N, hipStream_t s, bool writes_planes,
                            SideScope* side) {
    // side: the weight-gradient launches (they wait for decoder_bwd_kernel's row arrays only) go to the device's side stream
    // writes_planes: this call's plane gradient is built by plane_gather_kernel and WRITTEN into `gplanes` (the caller runs it before
    // the calls that add with atomics and has checked dec_bwd_writes_planes); otherwise it is added to what `gplanes` holds
    const long long P = (long long)B * N;
    if (P <= 0 || (head_mask & 15) == 0) return 0;
    const PackOff ko = pack_offsets();
    const BwdPackOff bo = bwd_pack_offsets();
    const ParamOff po = param_offsets(head_present);
    DecBwdArgs a{};
    a.planes = planes; a.p = p; a.blob = blob; a.bwd_blob = bwd_blob; a.gplanes = gplanes;
    a.B = B; a.N = N; a.P = P; a.invN = 1.0f / (float)N;
    float* sc = scratch;
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_mask >> h & 1)) continue;
        a.head_id[a.nheads] = h; a.head_off[a.nheads] = ko.dec32[h]; a.bwd_off[a.nheads] = bo.dec[h];
        a.out[a.nheads] = outs[h]; a.dout[a.nheads] = douts[h];
        a.scratch[a.nheads] = sc; sc += (size_t)P * 32 * DB_NARR;
        ++a.nheads;
    }
    a.Cbuf = sc; a.Pbuf = sc + (size_t)P * 96;
    a.dcbuf = gplanes && writes_planes && dec_bwd_writes_planes(a.nheads, B, N) ? a.Pbuf + (size_t)P * 32 : nullptr;
    const long long tiles = (P + 31) / 32;
    a.nbatch = (int)((tiles + 3) / 4);
    const int grid = a.nbatch < 256 ? a.nbatch : 256;
    const bool split = grid * a.nheads <= 256;                  // the point batches alone cannot fill the chip
    a.heads_per_wg = split ? 1 : a.nheads;
    const size_t lds = DEC32_BYTES > DECB_BYTES ? DEC32_BYTES : DECB_BYTES;
    giga::dyn_lds_once(reinterpret_cast<const void*>(decoder_bwd_kernel), (int)lds);
    GIGA_LAUNCH(decoder_bwd_kernel, dim3(grid, split ? a.nheads : 1), dim3(256), lds, s, a);
    if (a.dcbuf) {
        giga::dyn_lds_once(reinterpret_cast<const void*>(plane_gather_kernel), (int)PS_LDS);
        GIGA_LAUNCH(plane_gather_kernel, dim3(2, 3, B), dim3(PS_NW * 64), PS_LDS, s, a.dcbuf, p, gplanes, B, N);
    }
    // weight / bias gradients: one launch per head, or ONE for all heads when their problems fit one argument struct (the three
    // grasp heads of a one-query call: three 6-us launches of a handful of workgroups each)
    const bool merged = a.nheads > 1 && 17 * a.nheads <= LIN_MAX;
    int frc = 0;
    if (side) { frc = side->fork(); s = side->stream(); }
    LinArgs L{};
    auto flush = [&]() {
        L.nb_start[L.nprob] = L.nb_total;
        L.P = P;
        int ksplit = 2048 / L.nb_total;
        if (ksplit < 1) ksplit = 1;
        long long ppb = (P + ksplit - 1) / ksplit;
        ppb = (ppb + 31) / 32 * 32;
        L.pts_per_block = (int)ppb;
        ksplit = (int)((P + ppb - 1) / ppb);
        L.ksplit = ksplit;
        GIGA_LAUNCH(linear_wgrad_kernel, dim3(L.nb_total * ksplit), dim3(256), 0, s, L);
        L = LinArgs{};
    };
    for (int hh = 0; hh < a.nheads; ++hh) {
        const int h = a.head_id[hh];
        const HeadParamOff& o = po.head[h];
        const size_t AS = (size_t)P * 32;
        float* S = a.scratch[hh];
        auto add = [&](const float* R, const float* C, int ncol, float* dW, float* db, int sM, int mvalid, int nvalid) {
            LinProblem& q = L.pr[L.nprob];
            q.R = R; q.C = C; q.ncol = ncol; q.dW = dW; q.db = db; q.sM = sM; q.mvalid = mvalid; q.nvalid = nvalid;
            L.nb_start[L.nprob] = L.nb_total;
            L.nb_total += ncol / 32;
            ++L.nprob;
        };
        add(S + 22 * AS, S + 21 * AS, 32, grads + o.out_w, grads + o.out_b, 32, HEAD_OUT[h], 32);       // fc_out
        for (int i = 0; i < NBLK; ++i) {
            add(S + (i + 1) * AS, S + (16 + i) * AS, 32, grads + o.fc1_w[i], grads + o.fc1_b[i], 32, 32, 32);   // fc_1: DN[i+1], XH[i]
            add(S + (6 + i) * AS, S + (11 + i) * AS, 32, grads + o.fc0_w[i], grads + o.fc0_b[i], 32, 32, 32);   // fc_0: DH[i], XN[i]
            add(S + i * AS, a.Cbuf, 96, grads + o.fc_c_w[i], grads + o.fc_c_b[i], 96, 32, 96);                   // fc_c: DN[i], C
        }
        add(S + 0 * AS, a.Pbuf, 32, grads + o.fc_p_w, grads + o.fc_p_b, 3, 32, 3);                             // fc_p: DN[0], p
        if (!merged) flush();
    }
    if (merged) flush();
    return hipGetLastError() == hipSuccess ? frc : -10;
}

}  // namespace giga
