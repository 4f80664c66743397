// Fused training loss of the joint GIGA objective and its gradient with respect to the four head outputs.
//
// Replaces the caller-side torch code of the reference's training step for the literal call shape (one grasp query per
// scene): `select` (scripts/train_giga.py:154-158: squeeze + sigmoid(occ)) followed by `loss_fn` (train_giga.py:161-195):
//   loss_qual  = BCE(qual, label)                                  (:177-178; F.binary_cross_entropy clamps log at -100)
//   loss_rot   = min_k (1 - |<rot, rotations[:, k]>|), k = 0, 1    (:181-188)
//   loss_width = (40 width - 40 width_t)^2                         (:191-192, F.mse_loss(40 w, 40 w_t, 'none'))
//   loss_occ   = mean_M BCE(sigmoid(occ_logit), occ_t)             (:194-195)
//   loss       = mean_B (loss_qual + label * (loss_rot + 0.01 loss_width) + loss_occ)
// As ~20 tiny ATen kernels forward and as many backward that is a visible part of a 3 ms training step; here it is
// three launches: per-scene losses (one workgroup per scene, block reduction over the M occupancy queries), a fixed-order
// mean over the scenes (deterministic), and the gradient kernel.  HBM-bound elementwise work: 8 B read + 4 B written
// per occupancy query.  The arithmetic follows ATen's fp32 formulas (sigmoid as 1/(1+exp(-z)), logs clamped at -100,
// the BCE gradient (p - y) / max(p (1 - p), 1e-12)) so that losses and gradients agree with autograd through the
// reference's helpers to fp32 rounding (tests/test_gpu_training.py, golden G11).
#include <hip/hip_runtime.h>
#include "giga_launch.h"
#include <cmath>

namespace giga {

__device__ __forceinline__ float bce_term(float p, float y) {
    const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(logf(1.0f - p), -100.0f);
    return -(y * lp + (1.0f - y) * lq);
}
__device__ __forceinline__ float bce_grad(float p, float y) {        // d BCE / d p   (ATen binary_cross_entropy_backward)
    return (p - y) / fmaxf((1.0f - p) * p, 1e-12f);
}

// scene_loss [B][5]: qual, rot, width, occ, all
__global__ __launch_bounds__(256) void loss_scene_kernel(const float* __restrict__ qual, const float* __restrict__ rot,
                                                         const float* __restrict__ width, const float* __restrict__ occ,
                                                         const float* __restrict__ label, const float* __restrict__ rot_t,
                                                         const float* __restrict__ width_t, const float* __restrict__ occ_t,
                                                         int M, float* __restrict__ scene_loss) {
    __shared__ float part[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < M; i += 256) {
        const float z = occ[(size_t)b * M + i];
        const float p = 1.0f / (1.0f + expf(-z));
        s += bce_term(p, occ_t[(size_t)b * M + i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float locc = M > 0 ? ((part[0] + part[1]) + (part[2] + part[3])) / (float)M : 0.f;
        const float y = label[b];
        const float lq = bce_term(qual[b], y);
        float lr = 0.f;
        {
            const float* r = rot + 4 * b;
            const float* t0 = rot_t + 8 * b;
            const float d0 = r[0] * t0[0] + r[1] * t0[1] + r[2] * t0[2] + r[3] * t0[3];
            const float d1 = r[0] * t0[4] + r[1] * t0[5] + r[2] * t0[6] + r[3] * t0[7];
            lr = fminf(1.0f - fabsf(d0), 1.0f - fabsf(d1));
        }
        const float dw = 40.0f * width[b] - 40.0f * width_t[b];
        const float lw = dw * dw;
        float* o = scene_loss + 5 * b;
        o[0] = lq; o[1] = lr; o[2] = lw; o[3] = locc;
        o[4] = lq + y * (lr + 0.01f * lw) + locc;
    }
}

// losses [5] = mean over the scenes, summed in scene order by one wave (fixed order: run-to-run identical)
__global__ __launch_bounds__(64) void loss_mean_kernel(const float* __restrict__ scene_loss, int B, float* __restrict__ losses) {
    const int k = threadIdx.x;
    if (k >= 5) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += scene_loss[5 * b + k];
    losses[k] = s / (float)B;
}

// gradients of `loss` (times the upstream scalar *gout) with respect to the head outputs
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ qual, const float* __restrict__ rot,
                                                        const float* __restrict__ width, const float* __restrict__ occ,
                                                        const float* __restrict__ label, const float* __restrict__ rot_t,
                                                        const float* __restrict__ width_t, const float* __restrict__ occ_t,
                                                        const float* __restrict__ gout, int B, int M,
                                                        float* __restrict__ dqual, float* __restrict__ drot,
                                                        float* __restrict__ dwidth, float* __restrict__ docc) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float g = gout[0] / (float)B;
    const float gm = M > 0 ? g / (float)M : 0.f;
    for (int i = tid; i < M; i += 256) {
        const float z = occ[(size_t)b * M + i];
        const float p = 1.0f / (1.0f + expf(-z));
        // autograd's chain: BCE'(p) * sigmoid'(z) = (p - y) / max(p(1-p), 1e-12) * p(1-p)
        docc[(size_t)b * M + i] = gm * bce_grad(p, occ_t[(size_t)b * M + i]) * ((1.0f - p) * p);
    }
    if (tid == 0) {
        const float y = label[b];
        dqual[b] = g * bce_grad(qual[b], y);
        const float* r = rot + 4 * b;
        const float* t0 = rot_t + 8 * b;
        const float d0 = r[0] * t0[0] + r[1] * t0[1] + r[2] * t0[2] + r[3] * t0[3];
        const float d1 = r[0] * t0[4] + r[1] * t0[5] + r[2] * t0[6] + r[3] * t0[7];
        const float l0 = 1.0f - fabsf(d0), l1 = 1.0f - fabsf(d1);
        // torch.min(a, b): the gradient goes to the smaller argument (split evenly on an exact tie)
        const float w0 = l0 < l1 ? 1.0f : (l0 == l1 ? 0.5f : 0.0f), w1 = 1.0f - w0;
        const float s0 = d0 > 0.f ? 1.0f : (d0 < 0.f ? -1.0f : 0.0f), s1 = d1 > 0.f ? 1.0f : (d1 < 0.f ? -1.0f : 0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) drot[4 * b + k] = -g * y * (w0 * s0 * t0[k] + w1 * s1 * t0[4 + k]);
        dwidth[b] = g * y * 0.01f * 2.0f * (40.0f * width[b] - 40.0f * width_t[b]) * 40.0f;
    }
}

int launch_train_loss(const float* qual, const float* rot, const float* width, const float* occ, const float* label,
                      const float* rot_t, const float* width_t, const float* occ_t, int B, int M, float* losses,
                      float* scene_loss, hipStream_t s) {
    GIGA_LAUNCH(loss_scene_kernel, dim3(B), dim3(256), 0, s, qual, rot, width, occ, label, rot_t, width_t, occ_t, M,
                       scene_loss);
    GIGA_LAUNCH(loss_mean_kernel, dim3(1), dim3(64), 0, s, scene_loss, B, losses);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int launch_train_loss_backward(const float* qual, const float* rot, const float* width, const float* occ,
                               const float* label, const float* rot_t, const float* width_t, const float* occ_t,
                               const float* gout, int B, int M, float* dqual, float* drot, float* dwidth, float* docc,
                               hipStream_t s) {
    GIGA_LAUNCH(loss_grad_kernel, dim3(B), dim3(256), 0, s, qual, rot, width, occ, label, rot_t, width_t, occ_t, gout,
                       B, M, dqual, drot, dwidth, docc);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}


// ---------------------------------------------------------------------------------------------------------------------
// Adam over ONE flat fp32 parameter buffer (the module's flattened parameters, 581 863 elements): torch.optim.Adam's update
// (scripts/train_giga.py:49 builds `torch.optim.Adam(net.parameters(), lr)`; no amsgrad, not maximize) in one launch that
// uses the whole chip.  torch's fused multi-tensor kernel gives one tensor one workgroup per 65 536 elements: nine workgroups
// for this buffer, 98 us per step (tools/gpu_train_kernels.py); this takes ~5 us.  Same formulas in the same order as
// torch/optim/adam.py (_single_tensor_adam): m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g g;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps); L2 weight decay adds wd * p to g first.
__global__ __launch_bounds__(256) void adam_flat_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                        float4* __restrict__ v, size_t n4, size_t n, float step_size, float b2,
                                                        float omb1, float omb2, float eps, float wd, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        if (wd != 0.f) gg = fmaf(wd, pp, gg);
        mm = mm + (gg - mm) * omb1;                            // exp_avg.lerp_(grad, 1 - beta1)
        vv = vv * b2 + omb2 * gg * gg;                         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp - step_size * (mm / denom);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
    };
    if (i < n4) {
        float4 P = p[i], M = m[i], V = v[i];
        const float4 G = g[i];
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        p[i] = P; m[i] = M; v[i] = V;
    } else if (i == n4) {                                      // the (n % 4) tail elements
        float* ps = reinterpret_cast<float*>(p); const float* gs = reinterpret_cast<const float*>(g);
        float* ms = reinterpret_cast<float*>(m); float* vs = reinterpret_cast<float*>(v);
        for (size_t k = 4 * n4; k < n; ++k) upd(ps[k], gs[k], ms[k], vs[k]);
    }
}

int launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, double wd,
                     int step, hipStream_t s) {
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);     // (hyper-parameters stay double on the host, as in torch)
    const size_t n4 = n / 4;
    GIGA_LAUNCH(adam_flat_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<float4*>(p),
                       reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n4, n,
                       (float)(lr / bc1), (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, (float)sqrt(bc2));
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

}  // namespace giga
