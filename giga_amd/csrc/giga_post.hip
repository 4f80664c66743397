// Grasp post-processing on the device: the host-side scipy stage that follows the network in the reference's
// planner (src/vgn/detection_implicit.py:115-143 process, :87-97 bound, :146-174 select), batched over scenes.
//
//   K1/K2  separable Gaussian along axes 0,1  (scipy.ndimage.gaussian_filter, mode="nearest", truncate 4)
//   K3     Gaussian along axis 2 + validity mask (2-iteration masked 6-neighbour binary dilation of
//          tsdf > out_th) + gripper-width gate + workspace bound; counts voxels >= threshold per scene
//   K4     LOW_TH / threshold / force_detection logic + max-filter NMS (mode "reflect") + compaction of the
//          surviving voxels with their score, quaternion and width
//
// HBM-bound byte/stencil work on 40^3 volumes (256 KB per field): every kernel is one coalesced pass with
// the stencil neighbourhood served from L2/L1; no MFMA.  The Gaussian follows scipy's arithmetic (double
// accumulation in the symmetric-pair order, rounding to float32 after every axis) so that the thresholded
// selection is reproducible against the reference.
#include <hip/hip_runtime.h>
#include "giga_launch.h"

#include "../../include/giga_hip.h"

namespace {

constexpr int MAX_RADIUS = 16;

struct PostArgs {
    const float* tsdf;
    const float* qual;
    const float* rot;
    const float* width;
    float* tmp_a;
    float* tmp_b;
    float* qual_out;
    int* counters;     // [B][2]: {#voxels >= threshold, #candidates}
    int* cand_index;
    float* cand_score;
    float* cand_rot;
    float* cand_width;
    int B, R, cap;
    int radius;
    double w[MAX_RADIUS + 1];      // w[j] = weight at distance j from the centre
    float min_width, max_width, out_th, low_th, threshold;
    int lim_x, lim_y, lim_z;
    int filter_size, force_detection;
};

__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// scipy NI_Correlate1D, symmetric filter: tmp = x[0] w[0]; for j = r..1: tmp += (x[-j] + x[j]) w[j]
template <int AXIS>
__device__ __forceinline__ float gauss_axis(const PostArgs& a, const float* __restrict__ src, int x, int y, int z) {
    const int R = a.R;
    const int c = AXIS == 0 ? x : (AXIS == 1 ? y : z);
    const int stride = AXIS == 0 ? R * R : (AXIS == 1 ? R : 1);
    const float* line = src + (x * R + y) * R + z - c * stride;
    double acc = __dmul_rn((double)line[c * stride], a.w[0]);
    for (int j = a.radius; j >= 1; --j) {
        const double lo = (double)line[clampi(c - j, R) * stride];
        const double hi = (double)line[clampi(c + j, R) * stride];
        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(lo, hi), a.w[j]));
    }
    return (float)acc;
}

template <int AXIS>
__global__ __launch_bounds__(256) void post_gauss_kernel(PostArgs a) {
    const int R = a.R, V = R * R * R;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int b = blockIdx.y;
    const int z = v % R, y = (v / R) % R, x = v / (R * R);
    const float* src = (AXIS == 0 ? a.qual : a.tmp_a) + (size_t)b * V;
    float* dst = (AXIS == 0 ? a.tmp_a : a.tmp_b) + (size_t)b * V;
    dst[v] = gauss_axis<AXIS>(a, src, x, y, z);
}

__device__ __forceinline__ bool outside_at(const PostArgs& a, const float* t, int x, int y, int z) {
    const int R = a.R;
    if ((unsigned)x >= (unsigned)R || (unsigned)y >= (unsigned)R || (unsigned)z >= (unsigned)R) return false;
    return t[(x * R + y) * R + z] > a.out_th;
}

// one masked dilation step applied to "outside" (border value 0): state after iteration 1 at (x,y,z)
__device__ __forceinline__ bool dil1_at(const PostArgs& a, const float* t, int x, int y, int z) {
    const int R = a.R;
    if ((unsigned)x >= (unsigned)R || (unsigned)y >= (unsigned)R || (unsigned)z >= (unsigned)R) return false;
    const float tv = t[(x * R + y) * R + z];
    const bool out = tv > a.out_th;
    const bool inside = (1e-3f < tv) && (tv < a.out_th);
    if (out || inside) return out;           // already set, or not allowed to change
    return outside_at(a, t, x - 1, y, z) || outside_at(a, t, x + 1, y, z) || outside_at(a, t, x, y - 1, z) ||
           outside_at(a, t, x, y + 1, z) || outside_at(a, t, x, y, z - 1) || outside_at(a, t, x, y, z + 1);
}

__global__ __launch_bounds__(256) void post_mask_kernel(PostArgs a) {
    const int R = a.R, V = R * R * R;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    bool hit = false;
    if (v < V) {
        const int z = v % R, y = (v / R) % R, x = v / (R * R);
        float q = gauss_axis<2>(a, a.tmp_b + (size_t)b * V, x, y, z);
        const float* t = a.tsdf + (size_t)b * V;
        const float tv = t[v];
        const bool inside = (1e-3f < tv) && (tv < a.out_th);
        bool valid = dil1_at(a, t, x, y, z);
        if (!valid && !inside)
            valid = dil1_at(a, t, x - 1, y, z) || dil1_at(a, t, x + 1, y, z) || dil1_at(a, t, x, y - 1, z) ||
                    dil1_at(a, t, x, y + 1, z) || dil1_at(a, t, x, y, z - 1) || dil1_at(a, t, x, y, z + 1);
        const float wv = a.width[(size_t)b * V + v];
        if (!valid) q = 0.f;
        if (wv < a.min_width || wv > a.max_width) q = 0.f;
        if (x < a.lim_x || x >= R - a.lim_x || y < a.lim_y || y >= R - a.lim_y || z < a.lim_z) q = 0.f;
        a.qual_out[(size_t)b * V + v] = q;
        hit = !(q < a.low_th) && (q >= a.threshold);
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.counters[2 * b], __popcll(m));
}

__device__ __forceinline__ int reflecti(int i, int n) {
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return i;
}

__global__ __launch_bounds__(256) void post_nms_kernel(PostArgs a) {
    const int R = a.R, V = R * R * R;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int b = blockIdx.y;
    const float* q = a.qual_out + (size_t)b * V;
    const bool best_only = a.force_detection && a.counters[2 * b] == 0;
    const float floor_th = best_only ? a.low_th : fmaxf(a.low_th, a.threshold);
    // values below the floor count as 0 (detection_implicit.py:148-153); scores are >= 0 after process()
    float s = q[v];
    if (s < a.low_th || (!best_only && s < a.threshold)) s = 0.f;
    if (s == 0.f) return;
    const int z = v % R, y = (v / R) % R, x = v / (R * R);
    const int lo = -(a.filter_size / 2), hi = a.filter_size - a.filter_size / 2 - 1;
    float mx = 0.f;
    for (int dx = lo; dx <= hi; ++dx) {
        const int xx = reflecti(x + dx, R);
        for (int dy = lo; dy <= hi; ++dy) {
            const int yy = reflecti(y + dy, R);
            for (int dz = lo; dz <= hi; ++dz) mx = fmaxf(mx, q[(xx * R + yy) * R + reflecti(z + dz, R)]);
        }
    }
    (void)floor_th;
    if (mx > s) return;      // a larger neighbour also passes the floor (mx > s >= floor), so it survives thresholding
    const int slot = atomicAdd(&a.counters[2 * b + 1], 1);
    if (slot < a.cap) {
        const size_t o = (size_t)b * a.cap + slot;
        a.cand_index[o] = v;
        a.cand_score[o] = s;
        a.cand_width[o] = a.width[(size_t)b * V + v];
        reinterpret_cast<float4*>(a.cand_rot)[o] = reinterpret_cast<const float4*>(a.rot)[(size_t)b * V + v];
    }
}

}  // namespace

extern "C" size_t giga_grasp_workspace_bytes(int B, int R) {
    if (B <= 0 || R <= 0) return 0;
    return (size_t)2 * B * R * R * R * sizeof(float);
}

extern "C" int giga_grasp_select(const float* tsdf, const float* qual, const float* rot, const float* width, int B,
                                 int R, const GigaGraspParams* prm, float* qual_out, int* counters, int cap,
                                 int* cand_index, float* cand_score, float* cand_rot, float* cand_width,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!tsdf || !qual || !rot || !width || !prm || !qual_out || !counters || !cand_index || !cand_score ||
        !cand_rot || !cand_width || !workspace)
        return -6;
    if (B <= 0 || R < 2 || R > 128 || cap <= 0 || prm->max_filter_size < 1 || prm->max_filter_size > 16) return -1;
    if (!(prm->gaussian_sigma > 0.0)) return -1;
    if (workspace_bytes < giga_grasp_workspace_bytes(B, R)) return -4;
    PostArgs a{};
    a.tsdf = tsdf; a.qual = qual; a.rot = rot; a.width = width;
    a.tmp_a = static_cast<float*>(workspace);
    a.tmp_b = a.tmp_a + (size_t)B * R * R * R;
    a.qual_out = qual_out; a.counters = counters; a.cand_index = cand_index; a.cand_score = cand_score;
    a.cand_rot = cand_rot; a.cand_width = cand_width;
    a.B = B; a.R = R; a.cap = cap;
    // scipy.ndimage._filters._gaussian_kernel1d: radius = int(truncate * sigma + 0.5), truncate = 4
    const double sigma = prm->gaussian_sigma;
    a.radius = (int)(4.0 * sigma + 0.5);
    if (a.radius > MAX_RADIUS) return -1;
    double sum = 0.0, phi[2 * MAX_RADIUS + 1];
    for (int i = -a.radius; i <= a.radius; ++i) { phi[i + a.radius] = exp(-0.5 / (sigma * sigma) * (double)(i * i)); }
    for (int i = 0; i <= 2 * a.radius; ++i) sum += phi[i];
    for (int j = 0; j <= a.radius; ++j) a.w[j] = phi[a.radius + j] / sum;
    a.min_width = prm->min_width; a.max_width = prm->max_width; a.out_th = prm->out_th; a.low_th = prm->low_th;
    a.threshold = prm->threshold;
    a.lim_x = prm->lim_x; a.lim_y = prm->lim_y; a.lim_z = prm->lim_z;
    a.filter_size = prm->max_filter_size; a.force_detection = prm->force_detection;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counters, 0, sizeof(int) * 2 * B, s) != hipSuccess) return -10;
    const dim3 grid((R * R * R + 255) / 256, B);
    GIGA_LAUNCH(post_gauss_kernel<0>, grid, dim3(256), 0, s, a);
    GIGA_LAUNCH(post_gauss_kernel<1>, grid, dim3(256), 0, s, a);
    GIGA_LAUNCH(post_mask_kernel, grid, dim3(256), 0, s, a);
    GIGA_LAUNCH(post_nms_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}
