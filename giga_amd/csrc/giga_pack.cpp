// Host-side weight packing: reference state-dict (flat fp32, key order) -> MFMA fragment blob.
// Cold path (once per checkpoint load, reference networks.py:21-35 `load_network`).
//
// K-slot conventions (must match the kernels; see DESIGN.md "MFMA operand conventions"):
//  * 32x32 MFMA with WEIGHTS as the A operand (row n = output feature) and ACTIVATIONS as the B
//    operand (col = point/pixel).  The contraction index may be permuted freely as long as the A
//    and B images agree, which is what lets a layer's D registers feed the next layer's B operand
//    without any cross-lane movement.
#include "giga_layout.h"
#include "giga_conv32_geom.h"
#include "giga_dect.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace giga {

typedef _Float16 half_t;

static inline half_t f2h(float x) { return (half_t)x; }   // round-to-nearest-even
// fp32 -> bf16 bits, round-to-nearest-even (what v_cvt_pk_bf16_f32 does for finite values)
static inline uint16_t f2bf(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// hidden-layer feature held in k-slot (hi, j) of f16 chunk c   (D regs r = 8c + j)
static inline int hid16(int c, int hi, int j) { return drow(8 * c + j, hi); }
// input feature (0..95, concat order xz,xy,yz) in k-slot (hi,j) of f16 feature chunk c (0..5)
static inline int feat16(int c, int hi, int j) { return (c / 2) * 32 + (c % 2) * 16 + 8 * hi + j; }
// input feature in k-slot hi of fp32 feature MFMA m (0..47)
static inline int feat32(int m, int hi) { return (m / 16) * 32 + 16 * hi + (m % 16); }

static void pack_head16(const float* P, const HeadParamOff& o, int out_dim, uint8_t* dst) {
    std::memset(dst, 0, DEC16_BYTES);
    half_t* f = reinterpret_cast<half_t*>(dst);
    auto frag = [&](int idx, int lane, int j) -> half_t& { return f[(size_t)idx * 512 + lane * 8 + j]; };
    auto aux = [&](int idx, const float* wp, const float* bias_a, const float* bias_b) {
        for (int n = 0; n < 32; ++n) {
            float b = (bias_a ? bias_a[n] : 0.f) + (bias_b ? bias_b[n] : 0.f);
            half_t bh = f2h(b), bl = f2h(b - (float)bh);
            for (int k = 0; k < 3; ++k) {
                float w = wp ? wp[n * 3 + k] : 0.f;
                half_t wh = f2h(w), wl = f2h(w - (float)wh);
                frag(idx, n, k) = wh;          // x p_hi
                frag(idx, n, 4 + k) = wh;      // x p_lo
                frag(idx, 32 + n, k) = wl;     // (hi=1 lanes) x p_hi
            }
            frag(idx, n, 3) = bh;              // x 1.0
            frag(idx, n, 7) = bl;              // x 1.0
        }
    };
    auto dense32 = [&](int idx0, const float* W, int rows) {   // (rows,32) weight, 2 chunks
        for (int c = 0; c < 2; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                int n = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 8; ++j)
                    frag(idx0 + c, lane, j) = n < rows ? f2h(W[n * 32 + hid16(c, hi, j)]) : (half_t)0;
            }
    };
    int idx = 0;
    for (int b = 0; b < NBLK; ++b) {
        const float* Wc = P + o.fc_c_w[b];
        for (int c = 0; c < 6; ++c, ++idx)
            for (int lane = 0; lane < 64; ++lane) {
                int n = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 8; ++j) frag(idx, lane, j) = f2h(Wc[n * 96 + feat16(c, hi, j)]);
            }
        if (b == 0) aux(idx, P + o.fc_p_w, P + o.fc_p_b, P + o.fc_c_b[0]);
        else aux(idx, nullptr, P + o.fc_c_b[b], P + o.fc1_b[b - 1]);
        ++idx;
        dense32(idx, P + o.fc0_w[b], 32); idx += 2;
        dense32(idx, P + o.fc1_w[b], 32); idx += 2;
    }
    aux(idx, nullptr, P + o.fc1_b[NBLK - 1], nullptr); ++idx;
    dense32(idx, P + o.out_w, out_dim); idx += 2;
    float* ctab = reinterpret_cast<float*>(dst + (size_t)DEC16_FRAGS * FRAG);
    for (int b = 0; b < NBLK; ++b)
        for (int n = 0; n < 32; ++n) ctab[b * 32 + n] = P[o.fc0_b[b] + n];
    for (int n = 0; n < 32; ++n) ctab[NBLK * 32 + n] = n < out_dim ? P[o.out_b + n] : 0.f;
}

// f16x3 split image (giga_layout.h DEC16S_*): every weight w is stored as the pair hi = f16(w), lo = f16(w - hi) in two
// consecutive fragments with the k-slot maps of pack_head16.  lo is subnormal in f16 for |w| < 2^-3; the f16 MFMA keeps
// subnormal inputs (profiles/r02a_f16_mfma_denormals.txt), so hi + lo carries w to ~2^-22 relative (2^-25 absolute).
static void pack_head16s(const float* P, const HeadParamOff& o, int out_dim, uint8_t* dst) {
    std::memset(dst, 0, DEC16S_BYTES);
    half_t* f = reinterpret_cast<half_t*>(dst);
    auto frag = [&](int idx, int lane, int j) -> half_t& { return f[(size_t)idx * 512 + lane * 8 + j]; };
    auto put = [&](int idx, int lane, int j, float w) {     // pair (idx, idx + 1)
        const half_t h = f2h(w);
        frag(idx, lane, j) = h;
        frag(idx + 1, lane, j) = f2h(w - (float)h);
    };
    auto aux = [&](int idx, const float* wp, const float* bias_a, const float* bias_b) {   // as pack_head16
        for (int n = 0; n < 32; ++n) {
            float b = (bias_a ? bias_a[n] : 0.f) + (bias_b ? bias_b[n] : 0.f);
            half_t bh = f2h(b), bl = f2h(b - (float)bh);
            for (int k = 0; k < 3; ++k) {
                float w = wp ? wp[n * 3 + k] : 0.f;
                half_t wh = f2h(w), wl = f2h(w - (float)wh);
                frag(idx, n, k) = wh;
                frag(idx, n, 4 + k) = wh;
                frag(idx, 32 + n, k) = wl;
            }
            frag(idx, n, 3) = bh;
            frag(idx, n, 7) = bl;
        }
    };
    auto dense32 = [&](int idx0, const float* W, int rows) {   // (rows,32) weight: 2 chunks x [hi, lo]
        for (int c = 0; c < 2; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                int n = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 8; ++j) put(idx0 + 2 * c, lane, j, n < rows ? W[n * 32 + hid16(c, hi, j)] : 0.f);
            }
    };
    int idx = 0;
    for (int b = 0; b < NBLK; ++b) {
        const float* Wc = P + o.fc_c_w[b];
        for (int c = 0; c < 6; ++c, idx += 2)
            for (int lane = 0; lane < 64; ++lane) {
                int n = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 8; ++j) put(idx, lane, j, Wc[n * 96 + feat16(c, hi, j)]);
            }
        if (b == 0) aux(idx, P + o.fc_p_w, P + o.fc_p_b, P + o.fc_c_b[0]);
        else aux(idx, nullptr, P + o.fc_c_b[b], P + o.fc1_b[b - 1]);
        ++idx;
        dense32(idx, P + o.fc0_w[b], 32); idx += 4;
        dense32(idx, P + o.fc1_w[b], 32); idx += 4;
    }
    aux(idx, nullptr, P + o.fc1_b[NBLK - 1], nullptr); ++idx;
    dense32(idx, P + o.out_w, out_dim); idx += 4;
    float* ctab = reinterpret_cast<float*>(dst + (size_t)DEC16S_FRAGS * FRAG);
    for (int b = 0; b < NBLK; ++b)
        for (int n = 0; n < 32; ++n) ctab[b * 32 + n] = P[o.fc0_b[b] + n];
    for (int n = 0; n < 32; ++n) ctab[NBLK * 32 + n] = n < out_dim ? P[o.out_b + n] : 0.f;
}

static void pack_head32(const float* P, const HeadParamOff& o, int out_dim, uint8_t* dst) {
    std::memset(dst, 0, DEC32_BYTES);
    float* f = reinterpret_cast<float*>(dst);
    auto frag = [&](int idx, int lane, int j) -> float& { return f[(size_t)idx * 256 + lane * 4 + j]; };
    // aux fragment (3 MFMAs): [px,py] [pz,1] [1,0]  x  [Wp0,Wp1] [Wp2,bias_a] [bias_b,0].  The two biases
    // ride in separate k-slots so that every fp32 word of the blob is a pure gather of ONE parameter
    // (giga_pack_map / the device-side repack of the training path rely on that).
    auto aux = [&](int idx, const float* wp, const float* bias_a, const float* bias_b) {
        for (int n = 0; n < 32; ++n) {
            frag(idx, n, 0) = wp ? wp[n * 3 + 0] : 0.f;        // MFMA0 slot0: px
            frag(idx, 32 + n, 0) = wp ? wp[n * 3 + 1] : 0.f;   // MFMA0 slot1: py
            frag(idx, n, 1) = wp ? wp[n * 3 + 2] : 0.f;        // MFMA1 slot0: pz
            frag(idx, 32 + n, 1) = bias_a ? bias_a[n] : 0.f;   // MFMA1 slot1: 1.0
            frag(idx, n, 2) = bias_b ? bias_b[n] : 0.f;        // MFMA2 slot0: 1.0   (slot1: 0)
        }
    };
    auto dense32 = [&](int idx0, const float* W, int rows) {   // 16 MFMAs = 4 frags
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                int n = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 4; ++j)
                    frag(idx0 + q, lane, j) = n < rows ? W[n * 32 + drow(4 * q + j, hi)] : 0.f;
            }
    };
    int idx = 0;
    for (int b = 0; b < NBLK; ++b) {
        const float* Wc = P + o.fc_c_w[b];
        for (int q = 0; q < 12; ++q, ++idx)
            for (int lane = 0; lane < 64; ++lane) {
                int n = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 4; ++j) frag(idx, lane, j) = Wc[n * 96 + feat32(4 * q + j, hi)];
            }
        if (b == 0) {
            // bias = fc_p.bias + fc_c[0].bias
            aux(idx, P + o.fc_p_w, P + o.fc_p_b, P + o.fc_c_b[0]);
        } else {
            aux(idx, nullptr, P + o.fc_c_b[b], P + o.fc1_b[b - 1]);
        }
        ++idx;
        dense32(idx, P + o.fc0_w[b], 32); idx += 4;
        dense32(idx, P + o.fc1_w[b], 32); idx += 4;
    }
    aux(idx, nullptr, P + o.fc1_b[NBLK - 1], nullptr); ++idx;
    dense32(idx, P + o.out_w, out_dim); idx += 4;
    float* ctab = reinterpret_cast<float*>(dst + (size_t)DEC32_FRAGS * FRAG);
    for (int b = 0; b < NBLK; ++b)
        for (int n = 0; n < 32; ++n) ctab[b * 32 + n] = P[o.fc0_b[b] + n];
    for (int n = 0; n < 32; ++n) ctab[NBLK * 32 + n] = n < out_dim ? P[o.out_b + n] : 0.f;
}

// weight element W[co][ci][tap] of conv layer l in reference layout
static inline float conv_w_at(const float* W, const ConvLayerDesc& d, int co, int ci, int tap) {
    const int cin = d.cin0 + d.cin1;
    if (d.kind == CONV3) return W[((size_t)co * cin + ci) * 9 + tap];           // (Cout,Cin,3,3)
    if (d.kind == UPCONV) return W[((size_t)ci * d.cout + co) * 4 + tap];       // (Cin,Cout,2,2)
    return W[(size_t)co * cin + ci];                                            // (Cout,Cin,1,1)
}

static void pack_conv(const float* P, const ParamOff& po, const PackOff& ko, int l, uint8_t* blob) {
    // 16x16 MFMA B-operand fragments: lane (j = lane&15 -> output channel, g = lane>>4 -> k-slot group)
    //   f32 (v_mfma_f32_16x16x4_f32, 4 per fragment): 4 floats e -> W[co = nb*16+j][ci = kg*16 + 4g + e]
    //   f16 (v_mfma_f32_16x16x32_f16, 1 per fragment): 8 halfs e -> W[co][ci = kg*32 + 8g + e]
    // fragment order: [sub (ConvTranspose tap)][nb16][tap][kg]
    const ConvLayerDesc& d = kConv[l];
    const float* W = P + po.conv_w[l];
    const int cin = d.cin0 + d.cin1, taps = conv_taps(d), nsub = conv_nsub(d), nb16 = d.cout / 16;
    half_t* f16 = reinterpret_cast<half_t*>(blob + ko.conv[l].w16);
    half_t* f16s = reinterpret_cast<half_t*>(blob + ko.conv[l].w16s);      // split: fragment pair (2*i16, 2*i16 + 1) = (hi, lo)
    uint16_t* fbf = reinterpret_cast<uint16_t*>(blob + ko.conv[l].wbf);    // bf16, same fragment layout as f16
    float* f32 = reinterpret_cast<float*>(blob + ko.conv[l].w32);
    size_t i16 = 0, i32 = 0;
    for (int sub = 0; sub < nsub; ++sub)
        for (int nb = 0; nb < nb16; ++nb)
            for (int tap = 0; tap < taps; ++tap) {
                const int wtap = d.kind == UPCONV ? sub : tap;
                for (int kg = 0; kg < cin / 32; ++kg, ++i16)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int j = lane & 15, g = lane >> 4;
                        for (int e = 0; e < 8; ++e) {
                            const float w = conv_w_at(W, d, nb * 16 + j, kg * 32 + 8 * g + e, wtap);
                            const half_t h = f2h(w);
                            f16[i16 * 512 + lane * 8 + e] = h;
                            f16s[(2 * i16) * 512 + lane * 8 + e] = h;
                            f16s[(2 * i16 + 1) * 512 + lane * 8 + e] = f2h(w - (float)h);
                            fbf[i16 * 512 + lane * 8 + e] = f2bf(w);
                        }
                    }
                for (int kg = 0; kg < cin / 16; ++kg, ++i32)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int j = lane & 15, g = lane >> 4;
                        for (int e = 0; e < 4; ++e)
                            f32[i32 * 256 + lane * 4 + e] =
                                conv_w_at(W, d, nb * 16 + j, kg * 16 + 4 * g + e, wtap);
                    }
            }
    float* bias = reinterpret_cast<float*>(blob + ko.conv[l].bias);
    for (int c = 0; c < d.cout; ++c) bias[c] = P[po.conv_b[l] + c];
    // conv32 images (giga_conv32_geom.h): A operand of v_mfma_f32_32x32x16_{f16,bf16}, lane (i = lane&31 -> row, hi = lane>>5),
    // 8 halfs e -> W[co = 32 cs + c32_row_cout(i)][ci = 16 kc + 8 hi + e][tap]; fragment order [sub][cs][tap][kc]
    half_t* c32h = reinterpret_cast<half_t*>(blob + ko.conv[l].c32h);
    half_t* c32s = reinterpret_cast<half_t*>(blob + ko.conv[l].c32s);     // pair (2 f, 2 f + 1) = (hi, lo)
    uint16_t* c32b = reinterpret_cast<uint16_t*>(blob + ko.conv[l].c32b);
    size_t f = 0;
    for (int sub = 0; sub < nsub; ++sub)
        for (int cs = 0; cs < d.cout / 32; ++cs)
            for (int tap = 0; tap < taps; ++tap) {
                const int wtap = d.kind == UPCONV ? sub : tap;
                for (int kc = 0; kc < cin / 16; ++kc, ++f)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, hi = lane >> 5;
                        for (int e = 0; e < 8; ++e) {
                            const float w = conv_w_at(W, d, 32 * cs + c32_row_cout(i), 16 * kc + 8 * hi + e, wtap);
                            const half_t h = f2h(w);
                            c32h[f * 512 + lane * 8 + e] = h;
                            c32s[(2 * f) * 512 + lane * 8 + e] = h;
                            c32s[(2 * f + 1) * 512 + lane * 8 + e] = f2h(w - (float)h);
                            c32b[f * 512 + lane * 8 + e] = f2bf(w);
                        }
                    }
            }
}

// Winograd F(2x2, 3x3) image of a 3x3 layer (giga_wino.h): U = G g G^T per (co, ci), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],
// accumulated in double and rounded to fp32 once.  Order [grp = co / 16][kpass][pos = 4 xi + nu][chunk of 16 ci][half][lane][2]:
// lane (j = lane & 15 -> co = 16 grp + j, g = lane >> 4), element e -> ci = 64 kpass + 16 chunk + 4 g + 2 half + e -- the A operand of
// v_mfma_f32_16x16x4_f32 for the two k-steps of a half-chunk, 8 bytes per lane (a conflict-free ds_read_b64).
static void pack_wino(const float* P, const ParamOff& po, const PackOff& ko, int l, uint8_t* blob) {
    const ConvLayerDesc& d = kConv[l];
    if (d.kind != CONV3) return;
    const float* W = P + po.conv_w[l];
    const int cin = d.cin0 + d.cin1, kp = cin > 64 ? cin / 64 : 1, cinp = cin / kp, nchunk = cinp / 16;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    float* out = reinterpret_cast<float*>(blob + ko.conv[l].wino);
    size_t at = 0;
    for (int grp = 0; grp < d.cout / 16; ++grp)
        for (int k = 0; k < kp; ++k)
            for (int pos = 0; pos < 16; ++pos)
                for (int cc = 0; cc < nchunk; ++cc)
                    for (int h = 0; h < 2; ++h)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 2; ++e, ++at) {
                                const int co = 16 * grp + (lane & 15), ci = cinp * k + 16 * cc + 4 * (lane >> 4) + 2 * h + e;
                                const int xi = pos >> 2, nu = pos & 3;
                                double u = 0.0;
                                for (int ky = 0; ky < 3; ++ky)
                                    for (int kx = 0; kx < 3; ++kx)
                                        u += G[xi][ky] * G[nu][kx] * (double)conv_w_at(W, d, co, ci, ky * 3 + kx);
                                out[at] = (float)u;
                            }
}

static void write_stamp(uint8_t* at, int backward, size_t total) {
    PackStamp st{};
    std::memcpy(st.magic, "GIGAPACK", 8);
    st.abi_version = PACK_ABI_VERSION; st.backward = backward; st.total = total;
    std::memcpy(at, &st, sizeof st);
}
// 0: a blob of this library's layout; -8 otherwise
int packed_check_host(const uint8_t* blob, size_t bytes, int backward) {
    const size_t total = backward ? bwd_pack_offsets().total : pack_offsets().total;
    const size_t stamp = backward ? bwd_pack_offsets().stamp : pack_offsets().stamp;
    if (bytes != total) return -8;
    PackStamp st;
    std::memcpy(&st, blob + stamp, sizeof st);
    return std::memcmp(st.magic, "GIGAPACK", 8) == 0 && st.abi_version == PACK_ABI_VERSION && st.backward == backward && st.total == total ? 0 : -8;
}

size_t packed_bytes() { return pack_offsets().total; }

// returns 0 on success
int pack_weights_host(const float* P, size_t n_params, int head_present, uint8_t* blob, size_t blob_bytes) {
    const ParamOff po = param_offsets(head_present);
    const PackOff ko = pack_offsets();
    if (n_params != po.total) return -2;
    if (blob_bytes < ko.total) return -3;
    std::memset(blob, 0, ko.total);
    // conv_in: B operand of K-step s (v_mfma_f32_16x16x4_f32) for channel half h:
    //   [h][s][lane]: lane (j = lane&15, k = lane>>4) -> W[16h + j][tap = 4s + k], tap 27 = 0
    float* cw = reinterpret_cast<float*>(blob + ko.convin_w);
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < 7; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                int n = 16 * h + (lane & 15), tap = 4 * s + (lane >> 4);
                cw[(h * 7 + s) * 64 + lane] = tap < 27 ? P[po.conv_in_w + n * 27 + tap] : 0.f;
            }
    // f16x3 split conv_in: B operand of ONE v_mfma_f32_16x16x32_f16 per channel half (K = 27 taps padded to 32):
    //   [h][hi|lo][lane]: lane (j = lane&15 -> channel 16h + j, g = lane>>4), half e -> tap ci16_tap(g, e) (giga_layout.h: the
    //   slot order that makes the kernel's gather conflict-free; slots without a tap are zero)
    {
        half_t* cs = reinterpret_cast<half_t*>(blob + ko.convin_ws);
        for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * h + (lane & 15), tap = ci16_tap(lane >> 4, e);
                    const float w = tap >= 0 ? P[po.conv_in_w + n * 27 + tap] : 0.f;
                    const half_t hi = f2h(w);
                    cs[((2 * h) * 64 + lane) * 8 + e] = hi;
                    cs[((2 * h + 1) * 64 + lane) * 8 + e] = f2h(w - (float)hi);
                }
    }
    float* cb = reinterpret_cast<float*>(blob + ko.convin_b);
    for (int n = 0; n < 32; ++n) cb[n] = P[po.conv_in_b + n];
    for (int l = 0; l < NCONV; ++l) { pack_conv(P, po, ko, l, blob); pack_wino(P, po, ko, l, blob); }
    for (int h = 0; h < NHEADS; ++h) {
        if (!(head_present >> h & 1)) continue;
        pack_head16(P, po.head[h], HEAD_OUT[h], blob + ko.dec16[h]);
        pack_head32(P, po.head[h], HEAD_OUT[h], blob + ko.dec32[h]);
        pack_head16s(P, po.head[h], HEAD_OUT[h], blob + ko.dec16s[h]);
        dect_pack_fwd_host(blob + ko.dec32[h], blob + ko.dect[h]);        // bf16 training image, from the fp32 image just packed
    }
    // Folded variants.  The encoder ends with conv_final, a 1x1 convolution WITHOUT activation (unet.py:238), and the
    // decoder's first use of the planes is linear too: bilinear sampling (decoder.py:117-122) followed by fc_c (:169).
    // Sampling commutes with a per-pixel linear map, so  fc_c(sample(Wf x + bf)) = (Wc blockdiag(Wf,Wf,Wf)) sample(x) +
    // (bc + Wc [bf;bf;bf]):  with these fc_c weights the decoder reads the planes BEFORE conv_final and the encoder can
    // skip that layer (GIGA_FOLD_FINAL).  Products are accumulated in double, then rounded to fp32 once.
    {
        std::vector<float> Pf(P, P + n_params);
        const float* Wf = P + po.conv_w[NCONV - 1];          // [co][ci] (32 x 32 x 1 x 1)
        const float* bf = P + po.conv_b[NCONV - 1];
        for (int h = 0; h < NHEADS; ++h) {
            if (!(head_present >> h & 1)) continue;
            const HeadParamOff& ho = po.head[h];
            for (int b = 0; b < NBLK; ++b) {
                const float* Wc = P + ho.fc_c_w[b];           // [32][96]
                for (int n = 0; n < 32; ++n) {
                    double bacc = P[ho.fc_c_b[b] + n];
                    for (int pl = 0; pl < 3; ++pl) {
                        for (int ci = 0; ci < CD; ++ci) {
                            double acc = 0.0;
                            for (int co = 0; co < CD; ++co) acc += (double)Wc[n * 96 + pl * 32 + co] * Wf[co * CD + ci];
                            Pf[ho.fc_c_w[b] + n * 96 + pl * 32 + ci] = (float)acc;
                        }
                        for (int co = 0; co < CD; ++co) bacc += (double)Wc[n * 96 + pl * 32 + co] * bf[co];
                    }
                    Pf[ho.fc_c_b[b] + n] = (float)bacc;
                }
            }
            pack_head16(Pf.data(), ho, HEAD_OUT[h], blob + ko.dec16f[h]);
            pack_head32(Pf.data(), ho, HEAD_OUT[h], blob + ko.dec32f[h]);
            pack_head16s(Pf.data(), ho, HEAD_OUT[h], blob + ko.dec16sf[h]);
        }
    }
    write_stamp(blob + ko.stamp, 0, ko.total);
    return 0;
}

// ---- backward blob ----------------------------------------------------------------------------------------
static void pack_conv_dgrad(const float* P, const ParamOff& po, const BwdPackOff& bo, int l, uint8_t* blob) {
    const ConvLayerDesc& d = kConv[l];
    const float* W = P + po.conv_w[l];
    const int cin = d.cin0 + d.cin1;
    const int taps = d.kind == CONV3 ? 9 : d.kind == UPCONV ? 4 : 1;
    float* f32 = reinterpret_cast<float*>(blob + bo.conv[l]);
    // conv16 fragment order [nb16 (output channels' = forward ci)][tap'][kg (input channels' = forward co)]
    size_t i = 0;
    for (int nb = 0; nb < cin / 16; ++nb)
        for (int tap = 0; tap < taps; ++tap)
            for (int kg = 0; kg < d.cout / 16; ++kg, ++i)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 15, g = lane >> 4;
                    for (int e = 0; e < 4; ++e) {
                        const int ci = nb * 16 + j, co = kg * 16 + 4 * g + e;
                        const int wtap = d.kind == CONV3 ? 8 - tap : tap;
                        f32[i * 256 + lane * 4 + e] = conv_w_at(W, d, co, ci, wtap);
                    }
                }
}

// Winograd image of the DATA-GRADIENT convolution of a 3x3 layer (giga_wino.h): the convolution with cin' = cout, cout' = cin and
// W'[co'][ci'][tap'] = W[ci'][co'][8 - tap'] (pack_conv_dgrad), U' = G g' G^T summed over tap' in the order the device derive uses
// (giga_capi.hip::derive_wino_block on the backward blob's fp32 fragments), so both images are bit-identical.
static void pack_wino_dgrad(const float* P, const ParamOff& po, const BwdPackOff& bo, int l, uint8_t* blob) {
    const ConvLayerDesc& d = kConv[l];
    if (d.kind != CONV3) return;
    const float* W = P + po.conv_w[l];
    const int cinp_ = d.cout, coutp_ = d.cin0 + d.cin1;                   // the data-gradient convolution's channels in / out
    const int kp = cinp_ > 64 ? cinp_ / 64 : 1, cinp = cinp_ / kp, nchunk = cinp / 16;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    float* out = reinterpret_cast<float*>(blob + bo.wino[l]);
    size_t at = 0;
    for (int grp = 0; grp < coutp_ / 16; ++grp)
        for (int k = 0; k < kp; ++k)
            for (int pos = 0; pos < 16; ++pos)
                for (int cc = 0; cc < nchunk; ++cc)
                    for (int h = 0; h < 2; ++h)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 2; ++e, ++at) {
                                const int cop = 16 * grp + (lane & 15), cip = cinp * k + 16 * cc + 4 * (lane >> 4) + 2 * h + e;
                                const int xi = pos >> 2, nu = pos & 3;
                                double u = 0.0;
                                for (int ky = 0; ky < 3; ++ky)
                                    for (int kx = 0; kx < 3; ++kx)
                                        u += G[xi][ky] * G[nu][kx] * (double)conv_w_at(W, d, cip, cop, 8 - (ky * 3 + kx));
                                out[at] = (float)u;
                            }
}

// transposed decoder matrices for the backward chain (giga_decoder_bwd.hip):
//   fragment order per block b: Wc_b^T (3 row blocks of 32 input features x 4 frags), W0_b^T (4), W1_b^T (4)
//   A-operand fragment q of a transposed 32x32 matrix M^T: lane (i, hi), float j -> M[drow(4q+j, hi)][i]
//   then Wout as plain [4][32] floats (rows >= out_dim zero).
static void pack_head_bwd(const float* P, const HeadParamOff& o, int out_dim, uint8_t* dst) {
    std::memset(dst, 0, DECB_BYTES);
    float* f = reinterpret_cast<float*>(dst);
    auto frag = [&](int idx, int lane, int j) -> float& { return f[(size_t)idx * 256 + lane * 4 + j]; };
    auto transposed = [&](int idx0, const float* M, int ld, int col0) {   // M (32 x ld) row-major; columns col0..col0+31
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, hi = lane >> 5;
                for (int j = 0; j < 4; ++j) frag(idx0 + q, lane, j) = M[drow(4 * q + j, hi) * ld + col0 + i];
            }
    };
    int idx = 0;
    for (int b = 0; b < NBLK; ++b) {
        for (int rb = 0; rb < 3; ++rb) { transposed(idx, P + o.fc_c_w[b], 96, rb * 32); idx += 4; }
        transposed(idx, P + o.fc0_w[b], 32, 0); idx += 4;
        transposed(idx, P + o.fc1_w[b], 32, 0); idx += 4;
    }
    float* wout = f + (size_t)DECB_FRAGS * 256;
    for (int r = 0; r < out_dim; ++r)
        for (int k = 0; k < 32; ++k) wout[r * 32 + k] = P[o.out_w + r * 32 + k];
}

size_t bwd_packed_bytes() { return bwd_pack_offsets().total; }

int pack_bwd_host(const float* P, size_t n_params, int head_present, uint8_t* blob, size_t blob_bytes) {
    const ParamOff po = param_offsets(head_present);
    const BwdPackOff bo = bwd_pack_offsets();
    if (n_params != po.total) return -2;
    if (blob_bytes < bo.total) return -3;
    std::memset(blob, 0, bo.total);
    for (int l = 0; l < NCONV; ++l) { pack_conv_dgrad(P, po, bo, l, blob); pack_wino_dgrad(P, po, bo, l, blob); }
    for (int h = 0; h < NHEADS; ++h)
        if (head_present >> h & 1) {
            pack_head_bwd(P, po.head[h], HEAD_OUT[h], blob + bo.dec[h]);
            dect_pack_bwd_host(blob + bo.dec[h], blob + bo.dect[h]);
        }
    // bf16 images of the dgrad fragments: f16-layout fragment i16 (k-group of 32) = fp32 fragments 2*i16 and 2*i16 + 1
    for (int l = 0; l < NCONV; ++l) {
        const float* f32 = reinterpret_cast<const float*>(blob + bo.conv[l]);
        uint16_t* bf = reinterpret_cast<uint16_t*>(blob + bo.convbf[l]);
        for (int i16 = 0; i16 < bo.nfrag[l] / 2; ++i16)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int j = lane & 15, c = 8 * (lane >> 4) + e;
                    bf[(size_t)i16 * 512 + lane * 8 + e] =
                        f2bf(f32[((size_t)(2 * i16 + c / 16) * 64 + ((c % 16) / 4) * 16 + j) * 4 + c % 4]);
                }
    }
    write_stamp(blob + bo.stamp, 1, bo.total);
    return 0;
}

int pack_bwd_map_host(int head_present, int32_t* map, size_t nwords) {
    const ParamOff po = param_offsets(head_present);
    const BwdPackOff bo = bwd_pack_offsets();
    if (nwords < bo.total / 4) return -3;
    std::vector<float> probe(po.total);
    for (size_t i = 0; i < po.total; ++i) probe[i] = (float)(i + 1);
    std::vector<uint8_t> blob(bo.total);
    int rc = pack_bwd_host(probe.data(), po.total, head_present, blob.data(), blob.size());
    if (rc) return rc;
    const float* f = reinterpret_cast<const float*>(blob.data());
    for (size_t w = 0; w < bo.total / 4; ++w) map[w] = f[w] == 0.f ? -1 : (int32_t)f[w] - 1;
    for (size_t w = bo.convbf[0] / 4; w < bo.total / 4; ++w) map[w] = -2;      // bf16 images: derived on the device, not gathered
    return 0;
}

// ---- gather map for the device-side repack (training path: weights change every step) -----------------
// map[w] for every 4-byte word w of the blob:  >= 0 index of the fp32 parameter the word copies,
// -1 constant zero, -2 not an fp32 word (f16 fragments; left untouched by giga_repack_device).
// Built by probe-packing P[i] = i + 1 (exact in fp32 below 2^24): every fp32 word is a pure gather.
int pack_map_host(int head_present, int32_t* map, size_t nwords) {
    const ParamOff po = param_offsets(head_present);
    const PackOff ko = pack_offsets();
    if (nwords < ko.total / 4) return -3;
    if (po.total >= (1u << 24)) return -1;
    std::vector<float> probe(po.total);
    for (size_t i = 0; i < po.total; ++i) probe[i] = (float)(i + 1);
    std::vector<uint8_t> blob(ko.total);
    int rc = pack_weights_host(probe.data(), po.total, head_present, blob.data(), blob.size());
    if (rc) return rc;
    for (size_t w = 0; w < ko.total / 4; ++w) map[w] = -2;
    const float* f = reinterpret_cast<const float*>(blob.data());
    auto region = [&](size_t off, size_t bytes) {
        for (size_t w = off / 4; w < (off + bytes) / 4; ++w) map[w] = f[w] == 0.f ? -1 : (int32_t)f[w] - 1;
    };
    region(ko.convin_w, 14 * 64 * sizeof(float));
    region(ko.convin_b, CD * sizeof(float));
    for (int l = 0; l < NCONV; ++l) {
        region(ko.conv[l].w32, (size_t)ko.conv[l].nfrag32 * FRAG);
        region(ko.conv[l].bias, kConv[l].cout * sizeof(float));
    }
    for (int h = 0; h < NHEADS; ++h)
        if (head_present >> h & 1) region(ko.dec32[h], DEC32_BYTES);
    return 0;
}

}  // namespace giga
