// extern "C" entry points of libgiga_hip.so (declared in include/giga_hip.h).
#include <hip/hip_runtime.h>
#include "giga_launch.h"

#include "../../include/giga_hip.h"
#include "giga_layout.h"
#include "giga_conv32_geom.h"
#include "giga_dect.h"
#include "giga_args.h"
#include "giga_side.h"

namespace giga {
// giga_pack.cpp
size_t packed_bytes();
int pack_weights_host(const float* P, size_t n_params, int head_present, uint8_t* blob, size_t blob_bytes);
int pack_map_host(int head_present, int32_t* map, size_t nwords);
size_t bwd_packed_bytes();
int pack_bwd_host(const float* P, size_t n_params, int head_present, uint8_t* blob, size_t blob_bytes);
int pack_bwd_map_host(int head_present, int32_t* map, size_t nwords);
int packed_check_host(const uint8_t* blob, size_t bytes, int backward);
// giga_encoder_bwd.hip / giga_decoder_bwd.hip
constexpr size_t ENC_BWD_SYNC_BYTES = 8192;        // behind BwdWs::total: the counters of the persistent data-gradient kernel (giga_bwd_mega.h)
void persistent_forget();
int launch_encoder_backward(const float* tsdf, const uint8_t* blob, const uint8_t* bwd_blob, const uint8_t* fws,
                            float* gplanes, uint8_t* gws, float* grads, int head_present, int B, hipStream_t s, bool bf16_convs, bool convin_mask,
                            SideScope& side);
size_t dec_bwd_scratch_floats(long long P, int nheads);
bool dec_bwd_writes_planes(int nheads, int B, int N);
int launch_decoder_backward(const float* planes, const float* p, const uint8_t* blob, const uint8_t* bwd_blob,
                            int head_mask, const float* const* outs, const float* const* douts, float* gplanes,
                            float* grads, int head_present, float* scratch, int B, int N, hipStream_t s, bool writes_planes,
                            SideScope* side);
int launch_plane_gather(const float* dcbuf, const float* p, float* gplanes, int B, int N, hipStream_t s);
// giga_decoder_train16.hip (bf16 decoder of the bf16 training step)
int launch_dect_forward(const float* planes, const float* p, const uint8_t* blob, int head_mask, float* const* outs, int B, int N,
                        int post, hipStream_t s);
size_t dect_partial_floats(long long P, int nheads);
int launch_dect_backward(const float* planes, const float* p, const uint8_t* blob, const uint8_t* bwd_blob, int head_mask,
                         const float* const* outs, const float* const* douts, float* gplanes, float* dcbuf, float* scratch, int B,
                         int N, hipStream_t s, DectPending* pend);
int launch_dect_reduce(const DectPending& pend, float* grads, int head_present, hipStream_t s);
// giga_loss.hip
int launch_train_loss(const float* qual, const float* rot, const float* width, const float* occ, const float* label,
                      const float* rot_t, const float* width_t, const float* occ_t, int B, int M, float* losses,
                      float* scene_loss, hipStream_t s);
int launch_train_loss_backward(const float* qual, const float* rot, const float* width, const float* occ,
                               const float* label, const float* rot_t, const float* width_t, const float* occ_t,
                               const float* gout, int B, int M, float* dqual, float* drot, float* dwidth, float* docc,
                               hipStream_t s);
// giga_tsdf.hip
int launch_tsdf_scatter(const int* index, const float* value, const int* offsets, int B, int R, int n, float* grid,
                        int* winner, hipStream_t s);
// giga_encoder.hip
EncWs enc_workspace(int B, int precision);
int launch_encoder(const float* tsdf, const uint8_t* blob, void* planes_nhwc, float* planes_nchw, int B,
                   int precision, uint8_t* ws, hipStream_t s, int probe_stage, void* ev0, void* ev1);
// giga_decoder.hip
int launch_decoder(const DecArgs& a, int precision, hipStream_t s, void* ev0, void* ev1);
int launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, double wd,
                     int step, hipStream_t s);
int launch_planes_pack(const float* xz, const float* xy, const float* yz, void* dst, int B, int precision, hipStream_t s);
int launch_lattice_resample(const void* planes, const float* lin, void* out, int B, int R, int precision, hipStream_t s,
                            bool planes_fp32 = false);
int launch_planes_unpack(const void* src, float* dst, int B, int precision, hipStream_t s);
}  // namespace giga

using namespace giga;

std::atomic<unsigned long long> giga::g_launch_count{0};
std::atomic<unsigned long long> giga::g_probe_target{0};
void* giga::g_probe_ev[2] = {nullptr, nullptr};
const char* volatile giga::g_probe_name = "";

// byte offset of head h's weight image for a precision (0 fp32, 1 f16, 2 f16x3 split), plain or with conv_final folded in
static size_t head_image_offset(const PackOff& ko, int h, int precision, bool fold) {
    if (precision == 2) return fold ? ko.dec16sf[h] : ko.dec16s[h];
    if (precision == 1) return fold ? ko.dec16f[h] : ko.dec16[h];
    return fold ? ko.dec32f[h] : ko.dec32[h];
}

// blob word w <- params[map[w]] (map >= 0) | 0 (map == -1) | untouched (map == -2)
__global__ void repack_kernel(const float* __restrict__ params, const int32_t* __restrict__ map,
                              float* __restrict__ words, size_t nwords) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    const int m = map[i];
    if (m >= 0) words[i] = params[m];
    else if (m == -1) words[i] = 0.f;
}

// bf16 fragment images (f16 fragment layout: lane (j, g) holds channels 8g..8g+7 of k-group kg32) from the fp32 fragments
// (lane (j, g') holds channels 4g'..4g'+3 of k-group kg16 = 2*kg32 + h): one workgroup per bf16 fragment.
struct BfRegions { size_t src[2 * NCONV], dst[2 * NCONV]; int first[2 * NCONV + 1]; int n; };
__device__ __forceinline__ void derive_bf16_block(uint8_t* fwd, uint8_t* bwd, const BfRegions& r, int blk) {
    int reg = 0;
    while (reg + 1 < r.n && blk >= r.first[reg + 1]) ++reg;
    const int i16 = blk - r.first[reg], lane = threadIdx.x;
    uint8_t* base = reg < NCONV ? fwd : bwd;
    if (!base) return;
    const float* f32 = reinterpret_cast<const float*>(base + r.src[reg]);
    __bf16* out = reinterpret_cast<__bf16*>(base + r.dst[reg]) + ((size_t)i16 * 64 + lane) * 8;
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * g + e;
        out[e] = (__bf16)f32[((size_t)(2 * i16 + c / 16) * 64 + ((c % 16) / 4) * 16 + j) * 4 + c % 4];
    }
}

// bf16 conv32 images (A operands of v_mfma_f32_32x32x16_bf16, giga_pack.cpp: lane (i, hi), element e -> W[32 cs + c32_row_cout(i)][16 kc +
// 8 hi + e][tap], fragment order [sub][cs][tap][kc]) from the fp32 conv16 fragments of the same layer (lane (j, g), element e ->
// W[16 nb + j][16 kg + 4 g + e][tap], order [sub][nb][tap][kg]): one workgroup per conv32 fragment.
struct C32bRegions { size_t src[NCONV], dst[NCONV]; int first[NCONV + 1]; int nbt[NCONV], taps[NCONV], kg[NCONV]; };
__device__ __forceinline__ void derive_c32b_block(uint8_t* fwd, const C32bRegions& r, int blk) {
    int l = 0;
    while (l + 1 < NCONV && blk >= r.first[l + 1]) ++l;
    const int f = blk - r.first[l], lane = threadIdx.x;
    const int KG = r.kg[l], TAPS = r.taps[l], NBT = r.nbt[l], CS = NBT / 2;
    const int kc = f % KG, tap = (f / KG) % TAPS, cs = (f / (KG * TAPS)) % CS, sub = f / (KG * TAPS * CS);
    const int i = lane & 31, hi = lane >> 5;
    const int co = 32 * cs + c32_row_cout(i), nb = co >> 4, j = co & 15;
    const float* f32 = reinterpret_cast<const float*>(fwd + r.src[l]);
    const size_t i32 = ((size_t)(sub * NBT + nb) * TAPS + tap) * KG + kc;
    __bf16* out = reinterpret_cast<__bf16*>(fwd + r.dst[l]) + ((size_t)f * 64 + lane) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int rr = 8 * hi + e;
        out[e] = (__bf16)f32[(i32 * 64 + (rr >> 2) * 16 + j) * 4 + (rr & 3)];
    }
}

// Winograd F(2x2, 3x3) images (giga_wino.h; giga_pack.cpp::pack_wino) from the fp32 conv16 fragments of the same layer: one workgroup per
// 512-byte piece [grp][kpass][pos][chunk][half] = 64 lanes x 2 floats, U = G g G^T accumulated in double in the packer's order, so the
// device image equals the host's bit for bit.
struct WinoRegions { size_t src[NCONV], dst[NCONV]; int first[NCONV + 1]; int cin[NCONV]; };
__device__ __forceinline__ void derive_wino_block(uint8_t* fwd, const WinoRegions& r, int blk) {
    int l = 0;
    while (l + 1 < NCONV && blk >= r.first[l + 1]) ++l;
    const int f = blk - r.first[l], lane = threadIdx.x;
    const int cin = r.cin[l], kp = cin > 64 ? cin / 64 : 1, cinp = cin / kp, nchunk = cinp / 16, KG = cin / 16;
    const int h = f & 1, cc = (f >> 1) % nchunk, pos = (f / (2 * nchunk)) & 15, k = (f / (32 * nchunk)) % kp, grp = f / (32 * nchunk * kp);
    const int j = lane & 15, g = lane >> 4, xi = pos >> 2, nu = pos & 3;
    const float* f32 = reinterpret_cast<const float*>(fwd + r.src[l]);           // [nb][tap][kg][lane (j, g')][e']: W[16 nb + j][16 kg + 4 g' + e'][tap]
    float* out = reinterpret_cast<float*>(fwd + r.dst[l]) + ((size_t)f * 64 + lane) * 2;
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ci = cinp * k + 16 * cc + 4 * g + 2 * h + e;
        double u = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const float w = f32[((((size_t)grp * 9 + ky * 3 + kx) * KG + ci / 16) * 64 + ((ci % 16) / 4) * 16 + j) * 4 + ci % 4];
                u += G[xi][ky] * G[nu][kx] * (double)w;
            }
        out[e] = (float)u;
    }
}

constexpr int DECT_DERIVE_BLOCKS = (59 + 51) * NHEADS;
struct DeriveArgs {
    BfRegions r; C32bRegions c; int n16, n32;
    size_t dec32_0, dec32_stride, dectf_0, dectf_stride, decb_0, decb_stride, dectb_0, dectb_stride;
    size_t convin_w, convin_ws;
};
__global__ __launch_bounds__(64) void derive_all_kernel(uint8_t* fwd, uint8_t* bwd, DeriveArgs d) {
    const int b = blockIdx.x;
    if (b < d.n16) derive_bf16_block(fwd, bwd, d.r, b);
    else if (b < d.n16 + d.n32) derive_c32b_block(fwd, d.c, b - d.n16);
    else if (b == d.n16 + d.n32 + DECT_DERIVE_BLOCKS) {
        // f16x3 split conv_in operands (giga_pack.cpp): [2 channel halves][hi | lo][lane (j, g)][8 halfs e] <- W[16 h + j][ci16_tap(g, e)],
        // from the fp32 image [h][K-step s of 4 taps][lane (j, k)] = W[16 h + j][4 s + k]
        if (!fwd) return;
        const float* cw = reinterpret_cast<const float*>(fwd + d.convin_w);
        _Float16* cs = reinterpret_cast<_Float16*>(fwd + d.convin_ws);
        const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) {
                const int tap = ci16_tap(g, e);
                const float w = tap >= 0 ? cw[(h * 7 + tap / 4) * 64 + (tap % 4) * 16 + j] : 0.f;
                const _Float16 hi = (_Float16)w;
                cs[((2 * h) * 64 + lane) * 8 + e] = hi;
                cs[((2 * h + 1) * 64 + lane) * 8 + e] = (_Float16)(w - (float)hi);
            }
    } else {
        const int k = b - d.n16 - d.n32;
        dect_derive_block(fwd, bwd, k % (59 + 51), k / (59 + 51), d.dec32_0, d.dec32_stride, d.dectf_0, d.dectf_stride, d.decb_0,
                          d.decb_stride, d.dectb_0, d.dectb_stride);
    }
}
__global__ void repack2_kernel(const float* __restrict__ params, const int32_t* __restrict__ map_a, float* __restrict__ words_a, size_t na,
                               const int32_t* __restrict__ map_b, float* __restrict__ words_b, size_t nb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t* map = map_a; float* words = words_a;
    if (i >= na) { i -= na; if (i >= nb) return; map = map_b; words = words_b; }
    const int m = map[i];
    if (m >= 0) words[i] = params[m];
    else if (m == -1) words[i] = 0.f;
}

// one launch for both blobs: blocks [0, nf) derive the forward blob's images, the rest the backward blob's
__global__ __launch_bounds__(64) void derive_wino_kernel(uint8_t* fwd, WinoRegions wf, int nf, uint8_t* bwd, WinoRegions wb) {
    const int b = (int)blockIdx.x;
    if (b < nf) derive_wino_block(fwd, wf, b);
    else derive_wino_block(bwd, wb, b - nf);
}

extern "C" {

/* Winograd images of the fp32 3x3 layers (giga_wino.h) from the fp32 fragments of the same DEVICE blob, after giga_repack_device:
 * the forward blob's (the layers themselves) and / or the backward blob's (their data-gradient convolutions: the same fragment layout
 * with the channels swapped) -- either pointer may be NULL */
int giga_derive_winograd(void* packed_dev, void* bwd_packed_dev, void* stream) {
    if (!packed_dev && !bwd_packed_dev) return -1;
    const PackOff ko = pack_offsets();
    const BwdPackOff bo = bwd_pack_offsets();
    WinoRegions w[2] = {};
    int n[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        if (!(pass ? bwd_packed_dev : packed_dev)) continue;
        int nw = 0;
        for (int l = 0; l < NCONV; ++l) {
            const ConvLayerDesc& cd = kConv[l];
            w[pass].src[l] = pass ? bo.conv[l] : ko.conv[l].w32; w[pass].dst[l] = pass ? bo.wino[l] : ko.conv[l].wino; w[pass].first[l] = nw;
            w[pass].cin[l] = pass ? cd.cout : cd.cin0 + cd.cin1;                      // channels IN of the convolution the image is for
            if (cd.kind == CONV3) nw += (cd.cin0 + cd.cin1) * cd.cout / 8;           // 512-byte pieces of the layer's image
        }
        w[pass].first[NCONV] = nw;
        n[pass] = nw;
    }
    GIGA_LAUNCH(derive_wino_kernel, dim3(n[0] + n[1]), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<uint8_t*>(packed_dev), w[0], n[0],
                static_cast<uint8_t*>(bwd_packed_dev), w[1]);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

int giga_derive_bf16_fragments(void* packed_dev, void* bwd_packed_dev, void* stream) {
    if (!packed_dev && !bwd_packed_dev) return -1;
    const PackOff ko = pack_offsets();
    const BwdPackOff bo = bwd_pack_offsets();
    DeriveArgs d{};
    int at = 0;
    for (int l = 0; l < NCONV; ++l) { d.r.src[l] = ko.conv[l].w32; d.r.dst[l] = ko.conv[l].wbf; d.r.first[l] = at; at += ko.conv[l].nfrag16; }
    for (int l = 0; l < NCONV; ++l) {
        d.r.src[NCONV + l] = bo.conv[l]; d.r.dst[NCONV + l] = bo.convbf[l]; d.r.first[NCONV + l] = at; at += bo.nfrag[l] / 2;
    }
    d.r.first[2 * NCONV] = at;
    d.r.n = 2 * NCONV;
    d.n16 = at;
    int n32 = 0;
    for (int l = 0; l < NCONV; ++l) {
        const ConvLayerDesc& cd = kConv[l];
        d.c.src[l] = ko.conv[l].w32; d.c.dst[l] = ko.conv[l].c32b; d.c.first[l] = n32; n32 += ko.conv[l].nfragc32;
        d.c.nbt[l] = cd.cout / 16; d.c.taps[l] = conv_taps(cd); d.c.kg[l] = (cd.cin0 + cd.cin1) / 16;
    }
    d.c.first[NCONV] = n32;
    d.n32 = packed_dev ? n32 : 0;
    d.dec32_0 = ko.dec32[0]; d.dec32_stride = ko.dec32[1] - ko.dec32[0]; d.dectf_0 = ko.dect[0]; d.dectf_stride = ko.dect[1] - ko.dect[0];
    d.decb_0 = bo.dec[0]; d.decb_stride = bo.dec[1] - bo.dec[0]; d.dectb_0 = bo.dect[0]; d.dectb_stride = bo.dect[1] - bo.dect[0];
    d.convin_w = ko.convin_w; d.convin_ws = ko.convin_ws;
    // ONE launch: the bf16 conv16 fragments of both blobs, the bf16 conv32 images, the bf16 decoder images (giga_dect.h), the f16x3 conv_in operands
    GIGA_LAUNCH(derive_all_kernel, dim3(d.n16 + d.n32 + DECT_DERIVE_BLOCKS + 1), dim3(64), 0, static_cast<hipStream_t>(stream),
                static_cast<uint8_t*>(packed_dev), static_cast<uint8_t*>(bwd_packed_dev), d);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

/* both weight images in ONE launch (training: every step) */
int giga_repack_device2(const float* params_dev, const int32_t* map_fwd_dev, void* packed_dev, size_t nwords_fwd,
                        const int32_t* map_bwd_dev, void* bwd_packed_dev, size_t nwords_bwd, void* stream) {
    if (!params_dev || !map_fwd_dev || !packed_dev || !map_bwd_dev || !bwd_packed_dev) return -1;
    const size_t n = nwords_fwd + nwords_bwd;
    if (n == 0) return 0;
    GIGA_LAUNCH(repack2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), params_dev,
                map_fwd_dev, static_cast<float*>(packed_dev), nwords_fwd, map_bwd_dev, static_cast<float*>(bwd_packed_dev), nwords_bwd);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

static_assert(GIGA_ABI_VERSION == PACK_ABI_VERSION, "the blob stamp carries the ABI version");
int giga_abi_version(void) { return GIGA_ABI_VERSION; }

int giga_packed_check(const void* packed_host, size_t bytes, int backward) {
    if (!packed_host) return -1;
    return packed_check_host(static_cast<const uint8_t*>(packed_host), bytes, backward ? 1 : 0);
}

const char* giga_strerror(int code) {
    switch (code) {
        case 0: return "ok";
        case -1: return "invalid argument";
        case -2: return "parameter count does not match the head set";
        case -3: return "packed buffer too small";
        case -4: return "workspace too small";
        case -5: return "unsupported precision";
        case -6: return "null pointer for a requested output";
        case -7: return "more than GIGA_MAX_SCENES scenes in one call";
        case -8: return "packed blob was not produced by this version of the library (repack the weights)";
        case -10: return "HIP launch failed";
        default: return "unknown error";
    }
}

size_t giga_param_count(int head_present) { return param_offsets(head_present & 15).total; }

size_t giga_packed_bytes(void) { return packed_bytes(); }

int giga_pack_weights(const float* params_host, size_t n_params, int head_present, void* packed_host,
                      size_t packed_bytes_) {
    if (!params_host || !packed_host) return -1;
    return pack_weights_host(params_host, n_params, head_present & 15, static_cast<uint8_t*>(packed_host),
                             packed_bytes_);
}

int giga_pack_map(int head_present, int32_t* map_host, size_t nwords) {
    if (!map_host) return -1;
    return pack_map_host(head_present & 15, map_host, nwords);
}

int giga_repack_device(const float* params_dev, const int32_t* map_dev, void* packed_dev, size_t nwords,
                       void* stream) {
    if (!params_dev || !map_dev || !packed_dev) return -1;
    if (nwords == 0) return 0;
    GIGA_LAUNCH(repack_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), params_dev, map_dev, static_cast<float*>(packed_dev), nwords);
    return hipGetLastError() == hipSuccess ? 0 : -10;
}

size_t giga_encoder_workspace_bytes(int B, int precision) {
    precision &= ~(GIGA_FOLD_FINAL | GIGA_PERSIST_UNET | GIGA_LAYERWISE_UNET | GIGA_CONV32_UNET | GIGA_CONV16_UNET | GIGA_CONVIN_MASK | GIGA_DIRECT_CONV);
    if (B <= 0) return 0;
    return enc_workspace(B, precision).total;
}

int giga_encoder_workspace_layout(int B, int precision, size_t* offsets) {
    if (B <= 0 || !offsets) return -1;
    precision &= ~(GIGA_FOLD_FINAL | GIGA_PERSIST_UNET | GIGA_LAYERWISE_UNET | GIGA_CONV32_UNET | GIGA_CONV16_UNET | GIGA_CONVIN_MASK | GIGA_DIRECT_CONV);
    if (precision < 0 || precision > 3) return -5;
    const EncWs w = enc_workspace(B, precision);
    const size_t v[17] = {w.P0, w.A0, w.S0, w.Q0, w.A1, w.S1, w.Q1, w.A2, w.S2, w.U0, w.A3, w.A4, w.U1, w.A5, w.A6,
                          w.YZ, w.XZ};
    for (int i = 0; i < 17; ++i) offsets[i] = v[i];
    return 0;
}

int giga_encoder_forward_probe(const float* tsdf, const void* packed, void* planes_nhwc, float* planes_nchw,
                               int B, int precision, void* workspace, size_t workspace_bytes, void* stream,
                               int probe_stage, void* ev_start, void* ev_stop) {
    if (B < 0 || (B > 0 && (!tsdf || !packed || !planes_nhwc || !workspace))) return -1;
    if (B > GIGA_MAX_SCENES) return -7;
    const bool fold = (precision & GIGA_FOLD_FINAL) != 0;     // stop before conv_final (folded decoder images)
    const int persist = precision & (GIGA_PERSIST_UNET | GIGA_LAYERWISE_UNET | GIGA_CONV32_UNET | GIGA_CONV16_UNET | GIGA_CONVIN_MASK | GIGA_DIRECT_CONV);   // U-Net launch form and kernels (see the header)
    precision &= ~(GIGA_FOLD_FINAL | GIGA_PERSIST_UNET | GIGA_LAYERWISE_UNET | GIGA_CONV32_UNET | GIGA_CONV16_UNET | GIGA_CONVIN_MASK | GIGA_DIRECT_CONV);
    if (precision < 0 || precision > 3) return -5;
    // only the fp32 and bf16 encoders store conv_in's ReLU mask: a precision 1 / 2 forward that claims to would leave giga_backward
    // (GIGA_CONVIN_MASK_BWD) reading workspace bytes nobody wrote
    if ((persist & GIGA_CONVIN_MASK) && (precision == 1 || precision == 2)) return -5;
    if (fold && planes_nchw) return -1;                       // the reference-layout copy is the FINAL planes only
    if (workspace_bytes < giga_encoder_workspace_bytes(B, precision)) return -4;
    return launch_encoder(tsdf, static_cast<const uint8_t*>(packed), planes_nhwc, planes_nchw, B,
                          precision | (fold ? GIGA_FOLD_FINAL : 0) | persist,
                          static_cast<uint8_t*>(workspace), static_cast<hipStream_t>(stream), probe_stage,
                          ev_start, ev_stop);
}

int giga_encoder_forward(const float* tsdf, const void* packed, void* planes_nhwc, float* planes_nchw, int B,
                         int precision, void* workspace, size_t workspace_bytes, void* stream) {
    return giga_encoder_forward_probe(tsdf, packed, planes_nhwc, planes_nchw, B, precision, workspace,
                                      workspace_bytes, stream, -1, nullptr, nullptr);
}

unsigned long long giga_launch_count(void) { return g_launch_count.load(std::memory_order_relaxed); }

int giga_launch_probe(unsigned long long ordinal, void* ev_start, void* ev_stop) {
    if (ordinal != 0 && (!ev_start || !ev_stop)) return -1;
    g_probe_target.store(0, std::memory_order_relaxed);
    g_probe_ev[0] = ev_start; g_probe_ev[1] = ev_stop;
    g_probe_target.store(ordinal, std::memory_order_release);
    return 0;
}
const char* giga_launch_probe_name(void) { return g_probe_name; }

void giga_forget_device_state(void) {
    giga::dyn_lds_forget();
    giga::persistent_forget();
    giga::side_streams_forget();
}

void* giga_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? static_cast<void*>(e) : nullptr;
}
void giga_event_destroy(void* ev) { if (ev) (void)hipEventDestroy(static_cast<hipEvent_t>(ev)); }
int giga_event_record(void* ev, void* stream) {
    if (!ev) return -1;
    return hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -10;
}
int giga_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    if (!ev_start || !ev_stop || !ms) return -1;
    if (hipEventSynchronize(static_cast<hipEvent_t>(ev_stop)) != hipSuccess) return -10;
    return hipEventElapsedTime(ms, static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)) == hipSuccess
               ? 0 : -10;
}

int giga_planes_pack(const float* xz, const float* xy, const float* yz, void* planes_nhwc, int B, int precision,
                     void* stream) {
    if (B < 0 || (B > 0 && (!xz || !xy || !yz || !planes_nhwc))) return -1;
    if (precision < 0 || precision > 2) return -5;
    return launch_planes_pack(xz, xy, yz, planes_nhwc, B, precision, static_cast<hipStream_t>(stream));
}

int giga_planes_unpack(const void* planes_nhwc, float* planes_nchw, int B, int precision, void* stream) {
    if (B < 0 || (B > 0 && (!planes_nhwc || !planes_nchw))) return -1;
    if (precision < 0 || precision > 2) return -5;
    return launch_planes_unpack(planes_nhwc, planes_nchw, B, precision, static_cast<hipStream_t>(stream));
}

int giga_decoder_forward_probe(const void* planes_nhwc, const float* p, const void* packed, int head_mask,
                               float* qual, float* rot, float* width, float* occ, int B, int N, int precision,
                               int post, void* stream, void* ev_start, void* ev_stop);

int giga_decoder_forward(const void* planes_nhwc, const float* p, const void* packed, int head_mask, float* qual,
                         float* rot, float* width, float* occ, int B, int N, int precision, int post,
                         void* stream) {
    return giga_decoder_forward_probe(planes_nhwc, p, packed, head_mask, qual, rot, width, occ, B, N, precision,
                                      post, stream, nullptr, nullptr);
}

int giga_decoder_forward_probe(const void* planes_nhwc, const float* p, const void* packed, int head_mask,
                               float* qual, float* rot, float* width, float* occ, int B, int N, int precision,
                               int post, void* stream, void* ev_start, void* ev_stop) {
    if (B < 0 || N < 0) return -1;
    const bool fold = (precision & GIGA_FOLD_FINAL) != 0;     // planes are the encoder output BEFORE conv_final
    precision &= ~GIGA_FOLD_FINAL;
    if (precision < 0 || precision > 3 || (precision == 3 && fold)) return -5;
    if ((long long)B * N == 0 || (head_mask & 15) == 0) return 0;
    if (!planes_nhwc || !p || !packed) return -1;
    if (precision == 3) {                                     // bf16 operands / fp32 accumulate on fp32 planes: the bf16 training forward
        float* outs3[NHEADS] = {qual, rot, width, occ};
        for (int h = 0; h < NHEADS; ++h)
            if ((head_mask >> h & 1) && !outs3[h]) return -6;
        hipStream_t s3 = static_cast<hipStream_t>(stream);
        if (ev_start && hipEventRecord(static_cast<hipEvent_t>(ev_start), s3) != hipSuccess) return -10;
        const int rc3 = launch_dect_forward(static_cast<const float*>(planes_nhwc), p, static_cast<const uint8_t*>(packed), head_mask & 15,
                                            outs3, B, N, post, s3);
        if (ev_stop && hipEventRecord(static_cast<hipEvent_t>(ev_stop), s3) != hipSuccess) return -10;
        return rc3;
    }
    const PackOff ko = pack_offsets();
    DecArgs a{};
    a.planes = planes_nhwc; a.p = p; a.blob = static_cast<const uint8_t*>(packed);
    float* outs[NHEADS] = {qual, rot, width, occ};
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_mask >> h & 1)) continue;
        if (!outs[h]) return -6;
        a.head_id[a.nheads] = h;
        a.head_off[a.nheads] = head_image_offset(ko, h, precision, fold);
        a.out[a.nheads] = outs[h];
        ++a.nheads;
    }
    a.B = B; a.N = N; a.P = (long long)B * N; a.post = post;
    return launch_decoder(a, precision, static_cast<hipStream_t>(stream), ev_start, ev_stop);
}

size_t giga_lattice_workspace_bytes(int B, int R, int precision) {
    precision &= ~(GIGA_FOLD_FINAL | GIGA_PLANES_FP32);
    if (B <= 0 || R <= 0) return 0;
    return (size_t)3 * B * R * R * CD * (precision == 1 ? 2 : 4);
}

int giga_decoder_forward_lattice(const void* planes_nhwc, const float* lin, const void* packed, int head_mask,
                                 float* qual, float* rot, float* width, float* occ, int B, int R, int precision,
                                 int post, void* workspace, size_t workspace_bytes, void* stream, void* ev_start,
                                 void* ev_stop) {
    if (B < 0 || R < 0 || R > 64) return -1;
    const bool fold = (precision & GIGA_FOLD_FINAL) != 0;
    const bool planes32 = (precision & GIGA_PLANES_FP32) != 0;        // fp32 planes into the plain-f16 decoder
    precision &= ~(GIGA_FOLD_FINAL | GIGA_PLANES_FP32);
    if (precision < 0 || precision > 2 || (planes32 && precision != 1)) return -5;
    if (B == 0 || R == 0 || (head_mask & 15) == 0) return 0;
    if (!planes_nhwc || !lin || !packed || !workspace) return -1;
    if (workspace_bytes < giga_lattice_workspace_bytes(B, R, precision)) return -4;
    float* outs[NHEADS] = {qual, rot, width, occ};
    for (int h = 0; h < NHEADS; ++h)
        if ((head_mask >> h & 1) && !outs[h]) return -6;      // before anything is enqueued
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = launch_lattice_resample(planes_nhwc, lin, workspace, B, R, precision, s, planes32);
    if (rc) return rc;
    const PackOff ko = pack_offsets();
    DecArgs a{};
    a.planes = workspace; a.p = nullptr; a.blob = static_cast<const uint8_t*>(packed);
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_mask >> h & 1)) continue;
        a.head_id[a.nheads] = h;
        a.head_off[a.nheads] = head_image_offset(ko, h, precision, fold);
        a.out[a.nheads] = outs[h];
        ++a.nheads;
    }
    a.B = B; a.N = R * R * R; a.P = (long long)B * a.N; a.post = post; a.lin = lin; a.R = R;
    return launch_decoder(a, precision, s, ev_start, ev_stop);
}

// ---------------------------------------------------------------------------------------------------------
// training path (fp32)
size_t giga_bwd_packed_bytes(void) { return bwd_packed_bytes(); }

int giga_pack_bwd_weights(const float* params_host, size_t n_params, int head_present, void* packed_host,
                          size_t packed_bytes_) {
    if (!params_host || !packed_host) return -1;
    return pack_bwd_host(params_host, n_params, head_present & 15, static_cast<uint8_t*>(packed_host), packed_bytes_);
}

int giga_pack_bwd_map(int head_present, int32_t* map_host, size_t nwords) {
    if (!map_host) return -1;
    return pack_bwd_map_host(head_present & 15, map_host, nwords);
}

static size_t train_dec_scratch_bytes(int B, int N, int M, int head_present) {
    const int ng = __builtin_popcount(head_present & 7), nt = (head_present >> 3) & 1;
    const size_t a = ng ? dec_bwd_scratch_floats((long long)B * N, ng) : 0;
    const size_t b = nt && M > 0 ? dec_bwd_scratch_floats((long long)B * M, 1) : 0;
    // bf16 decoder (GIGA_BF16_DECODER): the dc rows of the occupancy head + the partial gradient tiles of both calls (they are
    // reduced together at the end); far below the fp32 path's row arrays except for a handful of points
    const size_t c = (nt && M > 0 ? (size_t)B * M * 96 + dect_partial_floats((long long)B * M, 1) : 0) +
                     (ng ? dect_partial_floats((long long)B * N, ng) : 0);
    const size_t m = a + b;                          // (both: the occupancy call's weight gradients run beside the grasp call, giga_side.h)
    return align_up((m > c ? m : c) * sizeof(float), 256);
}

size_t giga_backward_workspace_bytes(int B, int N, int M, int head_present) {
    if (B <= 0) return 0;
    head_present &= 15;
    return align_up((size_t)3 * B * RES * RES * CD * sizeof(float), 256) + enc_bwd_workspace(B).total + ENC_BWD_SYNC_BYTES +
           train_dec_scratch_bytes(B, N, M, head_present);
}

int giga_backward_workspace_layout(int B, size_t* offsets) {
    if (B <= 0 || !offsets) return -1;
    const BwdWs g = enc_bwd_workspace(B);
    const size_t base = align_up((size_t)3 * B * RES * RES * CD * sizeof(float), 256);     // the plane gradients come first
    const size_t o[15] = {g.gA6, g.gA5, g.gC1, g.gA4, g.gA3, g.gC0, g.gS2, g.gA2, g.gQ1, g.gS1, g.gA1, g.gQ0, g.gS0, g.gA0, g.gP0};
    for (int i = 0; i < 15; ++i) offsets[i] = base + o[i];
    return 0;
}

int giga_backward(const float* tsdf, const void* packed, const void* bwd_packed, const void* enc_workspace_fwd,
                  const void* planes_nhwc, const float* p, const float* p_tsdf, const float* const* outs,
                  const float* const* douts, float* grads, size_t n_params, int head_present, int B, int N, int M,
                  void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0) return 0;
    if (!tsdf || !packed || !bwd_packed || !enc_workspace_fwd || !planes_nhwc || !outs || !douts || !grads ||
        !workspace)
        return -1;
    if (B > GIGA_MAX_SCENES) return -7;
    const bool detach_occ = (head_present & GIGA_DETACH_OCC) != 0;     // detach_tsdf, models/__init__.py:61-63
    const bool bf16_convs = (head_present & GIGA_BF16_CONVS) != 0;     // data-gradient convolutions on bf16 MFMA
    const bool bf16_dec = (head_present & GIGA_BF16_DECODER) != 0;     // decoder backward on the fused bf16 kernel
    const bool convin_mask = (head_present & GIGA_CONVIN_MASK_BWD) != 0;   // the forward kept conv_in's ReLU mask (GIGA_CONVIN_MASK)
    head_present &= 15;
    if (N < 0 || M < 0) return -1;
    if (n_params != param_offsets(head_present).total) return -2;
    for (int h = 0; h < NHEADS; ++h) {                                  // every head that will run needs out and dout
        const bool runs = (head_present >> h & 1) && (h < 3 ? N > 0 : (M > 0 && p_tsdf != nullptr));
        if (runs && (!outs[h] || !douts[h])) return -6;
    }
    if ((head_present & 7) && N > 0 && !p) return -1;                   // (every argument is checked before anything is enqueued)
    if (workspace_bytes < giga_backward_workspace_bytes(B, N, M, head_present)) return -4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    const size_t gp_bytes = align_up((size_t)3 * B * RES * RES * CD * sizeof(float), 256);
    float* gplanes = reinterpret_cast<float*>(ws);
    uint8_t* gws = ws + gp_bytes;
    float* scratch = reinterpret_cast<float*>(gws + enc_bwd_workspace(B).total + ENC_BWD_SYNC_BYTES);
    const uint8_t* blob = static_cast<const uint8_t*>(packed);
    const uint8_t* bblob = static_cast<const uint8_t*>(bwd_packed);
    const bool occ_runs = (head_present & 8) && M > 0 && p_tsdf;
    // The occupancy head with many queries per scene WRITES the plane gradients (plane_gather_kernel: a gather over binned points
    // instead of global atomics), so it runs first and nothing needs clearing; the grasp heads then add theirs with atomics.
    const bool occ_writes = occ_runs && !detach_occ && dec_bwd_writes_planes(1, B, M);
    if (!occ_writes && hipMemsetAsync(gplanes, 0, gp_bytes, s) != hipSuccess) return -10;
    if (hipMemsetAsync(grads, 0, n_params * sizeof(float), s) != hipSuccess) return -10;
    int rc = 0;
    // work that waits for ONE launch of the caller's stream only goes to the device's side stream (giga_side.h; GIGA_WGRAD_STREAM=0: none)
    SideScope side(s, [] { const char* e = getenv("GIGA_WGRAD_STREAM"); return !e || atoi(e) != 0; }());
    if (bf16_dec) {
        // one fused launch per call (recompute, gradient chain, weight gradients), ONE reduce for every head of the step
        DectPending pend{};
        float* sc = scratch;
        if (occ_runs) {
            float* dcbuf = occ_writes ? sc : nullptr;
            if (dcbuf) sc += (size_t)B * M * 96;
            rc |= launch_dect_backward(static_cast<const float*>(planes_nhwc), p_tsdf, blob, bblob, 8, outs, douts,
                                       detach_occ ? nullptr : gplanes, dcbuf, sc, B, M, s, &pend);
            sc += dect_partial_floats((long long)B * M, 1);
            if (dcbuf) rc |= launch_plane_gather(dcbuf, p_tsdf, gplanes, B, M, s);
        }
        if ((head_present & 7) && N > 0)
            rc |= launch_dect_backward(static_cast<const float*>(planes_nhwc), p, blob, bblob, head_present & 7, outs, douts,
                                       gplanes, nullptr, sc, B, N, s, &pend);
        // (on the caller's stream: moved to the side stream it delays the first weight gradients there, 894 -> 899 us per step)
        rc |= launch_dect_reduce(pend, grads, head_present, s);
    } else {
    float* sc = scratch;
    if (occ_runs) {
        rc |= launch_decoder_backward(static_cast<const float*>(planes_nhwc), p_tsdf, blob, bblob, 8, outs, douts,
                                      detach_occ ? nullptr : gplanes, grads, head_present, sc, B, M, s, occ_writes, &side);
        sc += dec_bwd_scratch_floats((long long)B * M, 1);       // (its weight gradients may still be reading their rows)
    }
    if ((head_present & 7) && N > 0) {
        rc |= launch_decoder_backward(static_cast<const float*>(planes_nhwc), p, blob, bblob, head_present & 7, outs,
                                      douts, gplanes, grads, head_present, sc, B, N, s, false, &side);
    }
    }
    rc |= launch_encoder_backward(tsdf, blob, bblob, static_cast<const uint8_t*>(enc_workspace_fwd), gplanes, gws,
                                  grads, head_present, B, s, bf16_convs, convin_mask, side);
    return rc;
}

/* page-lock (and map for DMA) a host range the caller owns, e.g. a shared-memory batch ring written by reader processes */
int giga_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return -1;
    return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? 0 : -10;
}
int giga_host_unregister(void* ptr) {
    if (!ptr) return -1;
    return hipHostUnregister(ptr) == hipSuccess ? 0 : -10;
}

size_t giga_tsdf_scatter_workspace_bytes(int B, int R) {
    if (B <= 0 || R <= 0) return 0;
    return (size_t)B * R * R * R * sizeof(int32_t);
}

int giga_tsdf_scatter(const int32_t* voxel_index, const float* voxel_value, const int32_t* scene_offsets, int B, int R,
                      int n_voxels, float* grid, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || R <= 0 || R > 1024 || n_voxels < 0) return -1;
    if (B == 0) return 0;
    if (!grid || !scene_offsets) return -1;
    if (n_voxels > 0 && (!voxel_index || !voxel_value || !workspace)) return -1;
    if (n_voxels > 0 && workspace_bytes < giga_tsdf_scatter_workspace_bytes(B, R)) return -4;
    return launch_tsdf_scatter(voxel_index, voxel_value, scene_offsets, B, R, n_voxels, grid, static_cast<int*>(workspace),
                               static_cast<hipStream_t>(stream));
}

int giga_train_loss(const float* qual, const float* rot, const float* width, const float* occ_logits, const float* label,
                    const float* rot_targets, const float* width_target, const float* occ_target, int B, int M,
                    float* losses, float* scene_losses, void* stream) {
    if (B < 0 || M < 0) return -1;
    if (B == 0) return 0;
    if (!qual || !rot || !width || !label || !rot_targets || !width_target || !losses || !scene_losses) return -1;
    if (M > 0 && (!occ_logits || !occ_target)) return -1;
    return launch_train_loss(qual, rot, width, occ_logits, label, rot_targets, width_target, occ_target, B, M, losses,
                             scene_losses, static_cast<hipStream_t>(stream));
}

int giga_train_loss_backward(const float* qual, const float* rot, const float* width, const float* occ_logits,
                             const float* label, const float* rot_targets, const float* width_target,
                             const float* occ_target, const float* grad_loss, int B, int M, float* dqual, float* drot,
                             float* dwidth, float* docc, void* stream) {
    if (B < 0 || M < 0) return -1;
    if (B == 0) return 0;
    if (!qual || !rot || !width || !label || !rot_targets || !width_target || !grad_loss || !dqual || !drot || !dwidth)
        return -1;
    if (M > 0 && (!occ_logits || !occ_target || !docc)) return -1;
    return launch_train_loss_backward(qual, rot, width, occ_logits, label, rot_targets, width_target, occ_target, grad_loss,
                                      B, M, dqual, drot, dwidth, docc, static_cast<hipStream_t>(stream));
}

int giga_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, int step, void* stream) {
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return -1;
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0)) return -1;
    // the kernel moves float4: all four buffers must be 16-byte aligned (whole torch allocations are; a view at an odd offset is not)
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) return -1;
    return launch_adam_flat(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step,
                            static_cast<hipStream_t>(stream));
}

}  // extern "C"
