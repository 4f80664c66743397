// CPU emulation of the DATA MOVEMENT of giga_amd/csrc/giga_conv32.h (test infrastructure, not product code).
//
// The gfx950 kernel and this file share giga_conv32_geom.h: which LDS byte a staged value lands on, which LDS bytes a lane reads
// for (tile, tap, k-chunk), which output pixel a lane's accumulators belong to, which packed fragment a wave holds, how rows are
// dealt out to members, sub-bands, waves and tiles.  Here those functions drive a byte-accurate model of one layer -- an LDS image
// poisoned with NaN patterns before every sub-band, 64-lane operand images, the 32x32x16 MFMA's lane/register maps -- so that a
// wrong offset, a missing zero, a fragment-order mismatch with the packer or an output written twice / never shows up on the CPU,
// before a GPU minute is spent.  tests/test_conv32_emulation.py compares the result with torch's convolution.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "giga_conv32_geom.h"

using namespace giga;
typedef _Float16 half_t;

static inline float bf16_round(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float r;
    std::memcpy(&r, &u, 4);
    return r;
}
static inline float load16(const uint8_t* p, int mode) {        // one 16-bit operand element as float
    if (mode == C32_BF16) {
        uint32_t u = (uint32_t)(p[0] | (p[1] << 8)) << 16;
        float r;
        std::memcpy(&r, &u, 4);
        return r;
    }
    half_t h;
    std::memcpy(&h, p, 2);
    return (float)h;
}
static inline void store16(uint8_t* p, float v, int mode) {
    if (mode == C32_BF16) {
        const float r = bf16_round(v);
        uint32_t u;
        std::memcpy(&u, &r, 4);
        p[0] = (uint8_t)(u >> 16); p[1] = (uint8_t)(u >> 24);
    } else {
        const half_t h = (half_t)v;
        std::memcpy(p, &h, 2);
    }
}

template <class G, bool RELU>
static int emu_layer(const uint8_t* wimg, const float* bias, int Gimg, const float* in0, const float* in1, float* out,
                     float* out_pool, int* written, int* stats) {
    constexpr int MODE = G::MODE;
    std::vector<uint8_t> lds(C32_LDS);
    int max_lds = 0, ntiles_total = 0;
    for (int member = 0; member < C32_GROUP; ++member) {
        int sA, sB, nsb, rows;
        G::member_rows(member, Gimg, sA, sB);
        G::sub_bands(sA, sB, nsb, rows);
        const int sgm = G::member_sgm(member);
        for (int b = 0; b < nsb; ++b) {
            const int sb = sA + b * rows, R = (sB - sb) < rows ? (sB - sb) : rows;
            if (G::lds_bytes(R) > C32_LDS) return -1;
            if (G::lds_bytes(R) > max_lds) max_lds = G::lds_bytes(R);
            const int NT = G::n_tiles(R);
            ntiles_total += NT;
            const int ntb = G::batch_tiles(NT);
            const int nbatch = (NT + ntb - 1) / ntb;
            if (G::KP > 1 && nbatch > C32_NW) return -7;        // channel parts: one register tile per wave
            // accumulators of every (tile, member slice): bias, then the parts in order, taps x k-chunks inside a part
            std::vector<double> accs((size_t)NT * G::SPM * 64 * 16);
            std::vector<int> visits((size_t)NT * G::SPM, 0);
            for (int tt = 0; tt < NT; ++tt)
                for (int sl = 0; sl < G::SPM; ++sl)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 16; ++r)
                            accs[(((size_t)tt * G::SPM + sl) * 64 + lane) * 16 + r] = bias[32 * ((sgm * G::SPM + sl) % G::CS) + 16 * (lane >> 5) + r];
            for (int part = 0; part < G::KP; ++part) {
                std::memset(lds.data(), 0xFF, lds.size());      // NaN patterns: an unwritten byte that reaches a valid output is seen
                // ---- c32_fill: LDS fragment c <- blob fragment fill_src(c) ----
                for (int c = 0; c < G::WFR; ++c)
                    std::memcpy(lds.data() + (size_t)c * 1024, wimg + (size_t)G::fill_src(c, sgm, part) * 1024, 1024);
                // ---- c32_stage: thread t takes items t, t + NTHR, ... of the real rows; zero pixels separately ----
                int rrA, rrB;
                G::real_rows(sb, R, Gimg, rrA, rrB);
                const int nA = (rrB - rrA) * G::RI;
                std::vector<uint8_t> wrote(G::lds_bytes(R), 0);
                const size_t IMG = G::WBYTES;
                for (int tid = 0; tid < G::NTHR; ++tid) {
                    typename G::Cur k = G::cur_init(rrA, tid, sb);
                    const int ch0 = G::thr_ch(tid);
                    const bool first = G::KP > 1 ? G::part_tensor(part) == 0 : (G::C1 == 0 || ch0 < G::C0);
                    const float* src = first ? in0 : in1;
                    const int C = first ? G::C0 : G::C1;
                    const int ch = G::KP > 1 ? G::part_ch0(part) + ch0 : (first ? ch0 : ch0 - G::C0);
                    for (int j = tid; j < nA; j += G::NTHR, G::cur_next(k)) {
                        float x[8];
                        if (k.y < 0 || k.y >= G::H || k.x < 0 || k.x >= G::W || k.spix != k.rr * G::W + k.x || k.rr % G::H != k.y ||
                            k.rr < rrA || k.rr >= rrB || ch + 8 > C) return -2;
                        for (int e = 0; e < 8; ++e) {
                            float m = -INFINITY;
                            for (int q = 0; q < (G::POOLIN ? 4 : 1); ++q) {
                                const int pix = G::cur_src_pixel(k, q);
                                if (pix < 0 || pix >= Gimg * G::IH * G::IW) return -2;
                                m = std::fmax(m, src[(size_t)pix * C + ch + e]);
                            }
                            x[e] = m;
                        }
                        if (G::POOLIN && G::cur_own(k, sb, R) && out_pool)
                            for (int e = 0; e < 8; ++e) out_pool[(size_t)k.spix * G::C0 + ch0 + e] = x[e];
                        const int lo = k.lds;
                        if (lo < G::WBYTES || lo + G::ILB > G::lds_bytes(R)) return -3;
                        uint8_t* dst = lds.data() + lo;
                        for (int e = 0; e < G::ILB; ++e) wrote[lo + e] += 1;
                        for (int e = 0; e < 8; ++e) {
                            if (MODE == C32_SPLIT) {
                                const half_t h = (half_t)x[e];
                                const half_t l = (half_t)(x[e] - (float)h);
                                std::memcpy(dst + 2 * e, &h, 2);
                                std::memcpy(dst + 16 + 2 * e, &l, 2);
                            } else {
                                store16(dst + 2 * e, x[e], MODE);
                            }
                        }
                    }
                }
                if (G::HALO)
                    for (int q = 0; q < G::n_buf_pixels(R); ++q)
                        if (G::pad_pixel(q, sb, Gimg)) {
                            std::memset(lds.data() + IMG + (size_t)q * G::PS, 0, G::IPP * G::ILB);
                            for (int e = 0; e < G::IPP * G::ILB; ++e) wrote[IMG + (size_t)q * G::PS + e] += 1;
                        }
                for (int q = 0; q < G::n_buf_pixels(R); ++q)    // every channel byte of every buffer pixel exactly once
                    for (int e = 0; e < G::IPP * G::ILB; ++e)
                        if (wrote[IMG + (size_t)q * G::PS + e] != 1) return -6;
                // ---- tiles: batches of ntb tiles dealt over the eight waves; per batch all slices of the member ----
                for (int wave = 0; wave < C32_NW; ++wave)
                    for (int bi = wave; bi < nbatch; bi += C32_NW)
                        for (int j = 0; j < ntb; ++j) {
                            const int tt = bi * ntb + j;
                            if (tt >= NT) continue;
                            for (int sl = 0; sl < G::SPM; ++sl) {      // (pass sp = sl / SPW, slice s = sl % SPW of the register tile)
                                double* acc = &accs[(((size_t)tt * G::SPM + sl) * 64) * 16];
                                visits[(size_t)tt * G::SPM + sl] += 1;
                                for (int it = 0; it < G::NIT; ++it) {
                                    const int tap = it / G::KCP, kcp = it % G::KCP;
                                    float A[G::NOP][64][8], B[G::NOP][64][8];
                                    for (int o = 0; o < G::NOP; ++o)
                                        for (int lane = 0; lane < 64; ++lane) {
                                            const int off = G::lane_base(lane) + tt * G::tile_step() + G::tap_off(tap) + G::kc_off(kcp) + 16 * o;
                                            const int woff = lane * 16 + G::w_off(sl, it, o);      // = w_off(sp * SPW, 0, 0) + w_off(s, it, o)
                                            if (off < G::WBYTES || off + 16 > C32_LDS || woff + 16 > G::WBYTES) return -4;
                                            for (int e = 0; e < 8; ++e) {
                                                A[o][lane][e] = load16(lds.data() + woff + 2 * e, MODE);
                                                B[o][lane][e] = load16(lds.data() + off + 2 * e, MODE);
                                            }
                                        }
                                    // D[i][n] = sum_{hi, e} A[hi*32 + i][e] * B[hi*32 + n][e]; lane (n, hi') register r = row (r&3) + 8 (r>>2) + 4 hi'
                                    for (int lane = 0; lane < 64; ++lane) {
                                        const int n = lane & 31, hq = lane >> 5;
                                        for (int r = 0; r < 16; ++r) {
                                            const int row = drow(r, hq);
                                            double d = 0.0;
                                            for (int h = 0; h < 2; ++h)
                                                for (int e = 0; e < 8; ++e) {
                                                    if (MODE == C32_SPLIT)
                                                        d += (double)A[1][h * 32 + row][e] * B[0][h * 32 + n][e] +
                                                             (double)A[0][h * 32 + row][e] * B[1][h * 32 + n][e] +
                                                             (double)A[0][h * 32 + row][e] * B[0][h * 32 + n][e];
                                                    else
                                                        d += (double)A[0][h * 32 + row][e] * B[0][h * 32 + n][e];
                                                }
                                            acc[lane * 16 + r] += d;
                                        }
                                    }
                                }
                            }
                        }
            }
            // ---- epilogue ----
            for (int tt = 0; tt < NT; ++tt)
                for (int sl = 0; sl < G::SPM; ++sl) {
                    if (visits[(size_t)tt * G::SPM + sl] != G::KP) return -8;
                    const int slice = sgm * G::SPM + sl, sub = slice / G::CS, cs = slice % G::CS;
                    const double* acc = &accs[(((size_t)tt * G::SPM + sl) * 64) * 16];
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = lane & 31, hq = lane >> 5;
                        const typename G::Out o = G::out_pixel(tt, n, sb, R);
                        if (!o.valid) continue;
                        if (o.g < 0 || o.g >= Gimg || o.y >= G::H || o.x < 0) return -5;
                        const size_t pix = (size_t)G::out_index(o.g, o.y, o.x, sub);
                        for (int r = 0; r < 16; ++r) {
                            float v = (float)acc[lane * 16 + r];
                            if (RELU) v = v > 0.f ? v : 0.f;
                            if (MODE == C32_NATIVE) v = (float)(half_t)v;
                            const size_t idx = pix * G::COUT + 32 * cs + 16 * hq + r;
                            out[idx] = v;
                            written[idx] += 1;
                        }
                    }
                }
        }
    }
    if (stats) { stats[0] = max_lds; stats[1] = ntiles_total; stats[2] = G::RBMAX; stats[3] = G::PS; }
    return 0;
}

extern "C" {

// byte offsets of layer `layer`'s conv32 image for `mode` and of its bias inside the packed blob
int conv32_emu_offsets(int layer, int mode, size_t* w_off, size_t* bias_off) {
    const PackOff ko = pack_offsets();
    if (layer < 0 || layer >= NCONV) return -1;
    *w_off = mode == C32_SPLIT ? ko.conv[layer].c32s : mode == C32_BF16 ? ko.conv[layer].c32b : ko.conv[layer].c32h;
    *bias_off = ko.conv[layer].bias;
    return 0;
}

// in0 / in1: fp32 NHWC [G][IH][IW][C0 / C1] (values representable in the mode's storage type); out: [G][OH][OW][COUT];
// out_pool: [G][H][W][C0] for the POOLIN layers (may be NULL); written: one counter per output element.
int conv32_emu_layer(int layer, int mode, const uint8_t* blob, int Gimg, const float* in0, const float* in1, float* out,
                     float* out_pool, int* written, int* stats) {
    size_t w_off, b_off;
    if (conv32_emu_offsets(layer, mode, &w_off, &b_off)) return -10;
    const uint8_t* wimg = blob + w_off;
    const float* bias = reinterpret_cast<const float*>(blob + b_off);
#define X(l, KIND, C0, C1, COUT, H, W, POOLIN, SPW, SGN, SGS, KPS)                                                               \
    if (layer == l) {                                                                                                    \
        if (mode == C32_NATIVE) return emu_layer<U32Layer<C32_NATIVE, l>::G, KIND == CONV3>(wimg, bias, Gimg, in0, in1, out, out_pool, written, stats); \
        if (mode == C32_SPLIT) return emu_layer<U32Layer<C32_SPLIT, l>::G, KIND == CONV3>(wimg, bias, Gimg, in0, in1, out, out_pool, written, stats);   \
        if (mode == C32_BF16) return emu_layer<U32Layer<C32_BF16, l>::G, KIND == CONV3>(wimg, bias, Gimg, in0, in1, out, out_pool, written, stats);     \
    }
    GIGA_UNET32_LAYERS(X)
#undef X
    return -11;
}

}  // extern "C"
