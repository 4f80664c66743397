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
                // (KP == 1: (batch, slice pass) tasks dealt over the waves; KP > 1: one batch per wave, NSP == 1)
                for (int wave = 0; wave < C32_NW; ++wave)
                    for (int task = wave; task < nbatch * G::NSP; task += C32_NW)
                        for (int j = 0; j < ntb; ++j) {
                            const int bi = task / G::NSP, sp = task - bi * G::NSP;
                            const int tt = bi * ntb + j;
                            if (tt >= NT) continue;
                            for (int sl = sp * G::SPW; sl < (sp + 1) * G::SPW; ++sl) {      // the slices of the pass's register tile
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

// accumulate tile tt x member slice sl of layer G (all taps and k-chunks) from an LDS image: weights at byte wbase, staged image
// at wbase + G::WBYTES + ioff.  acc[lane * 16 + r].  Returns nonzero if an operand address leaves the allowed range.
template <class G>
static int tile_acc(const std::vector<uint8_t>& lds, int wbase, int ioff, int img_end, int tt, int sl, double* acc) {
    constexpr int MODE = G::MODE;
    for (int it = 0; it < G::NIT; ++it) {
        const int tap = it / G::KCP, kcp = it % G::KCP;
        float A[G::NOP][64][8], B[G::NOP][64][8];
        for (int o = 0; o < G::NOP; ++o)
            for (int lane = 0; lane < 64; ++lane) {
                const int off = wbase + ioff + G::lane_base(lane) + tt * G::tile_step() + G::tap_off(tap) + G::kc_off(kcp) + 16 * o;
                const int woff = lane * 16 + G::w_off(sl, it, o);
                if (off < wbase + ioff + G::WBYTES || off + 16 > img_end || woff + 16 > G::WBYTES) return -4;
                for (int e = 0; e < 8; ++e) {
                    A[o][lane][e] = load16(lds.data() + wbase + woff + 2 * e, MODE);
                    B[o][lane][e] = load16(lds.data() + off + 2 * e, MODE);
                }
            }
        for (int lane = 0; lane < 64; ++lane) {
            const int n = lane & 31, hq = lane >> 5;
            for (int r = 0; r < 16; ++r) {
                const int row = drow(r, hq);
                double d = 0.0;
                for (int h = 0; h < 2; ++h)
                    for (int e = 0; e < 8; ++e) {
                        if (MODE == C32_SPLIT)
                            d += (double)A[1][h * 32 + row][e] * B[0][h * 32 + n][e] + (double)A[0][h * 32 + row][e] * B[1][h * 32 + n][e] +
                                 (double)A[0][h * 32 + row][e] * B[0][h * 32 + n][e];
                        else
                            d += (double)A[0][h * 32 + row][e] * B[0][h * 32 + n][e];
                    }
                acc[lane * 16 + r] += d;
            }
        }
    }
    return 0;
}

// ---- a fused pair A -> B (C32Pair; giga_conv32.h: c32_run_pair) ------------------------------------------------------------------
// The member stages A's input for its rows with a halo of two, computes A for its rows + one above and below, writes those values
// into B's LDS image (B's format; B's zero rows / columns zeroed separately) and the rows it owns to memory, then computes B.
template <class GA, class GB, bool RELU_A, bool RELU_B>
static int emu_pair(const uint8_t* wA, const float* biasA, const uint8_t* wB, const float* biasB, int Gimg, const float* in0,
                    const float* in1, float* outA, float* out_pool, int* writtenA, float* outB, int* writtenB, int* stats) {
    using PR = C32Pair<GA, GB>;
    constexpr int MODE = GA::MODE;
    std::vector<uint8_t> lds(C32_LDS);
    int max_lds = 0, ntiles_total = 0;
    for (int member = 0; member < C32_GROUP; ++member) {
        int sA, sB, nsb, rows, sA2, sB2;
        GB::member_rows(member, Gimg, sA, sB);
        GA::member_rows(member, Gimg, sA2, sB2);
        if (sA2 != sA || sB2 != sB) return -20;                  // the rows of A a member stores are the rows it owns in B
        PR::sub_bands(sA, sB, nsb, rows);
        for (int b = 0; b < nsb; ++b) {
            const int sb = sA + b * rows, R = (sB - sb) < rows ? (sB - sb) : rows;
            if (R > PR::RBMAX || PR::lds_bytes(R) > C32_LDS) return -1;
            if (PR::lds_bytes(R) > max_lds) max_lds = PR::lds_bytes(R);
            std::memset(lds.data(), 0xFF, lds.size());
            for (int c = 0; c < GA::WFR; ++c) std::memcpy(lds.data() + (size_t)c * 1024, wA + (size_t)GA::fill_src(c, 0, 0) * 1024, 1024);
            for (int c = 0; c < GB::WFR; ++c)
                std::memcpy(lds.data() + PR::WA + (size_t)c * 1024, wB + (size_t)GB::fill_src(c, 0, 0) * 1024, 1024);
            // ---- stage A: sub-band (sb - 1, R + 2), image base = smem + WB ----
            const int sbA = sb - 1, RA = R + 2;
            int rrA, rrB;
            GA::real_rows(sbA, RA, Gimg, rrA, rrB);
            const int nA = (rrB - rrA) * GA::RI;
            std::vector<uint8_t> wrote(C32_LDS, 0);
            const int IMGA = PR::WA + PR::WB, MID = PR::mid0(R);
            if (IMGA + PR::imgA_bytes(R) != MID) return -21;
            for (int tid = 0; tid < GA::NTHR; ++tid) {
                typename GA::Cur k = GA::cur_init(rrA, tid, sbA);
                const int ch0 = GA::thr_ch(tid);
                const bool first = GA::C1 == 0 || ch0 < GA::C0;
                const float* src = first ? in0 : in1;
                const int C = first ? GA::C0 : GA::C1;
                const int ch = first ? ch0 : ch0 - GA::C0;
                for (int j = tid; j < nA; j += GA::NTHR, GA::cur_next(k)) {
                    float x[8];
                    if (k.y < 0 || k.y >= GA::H || k.x < 0 || k.x >= GA::W || k.spix != k.rr * GA::W + k.x || k.rr % GA::H != k.y ||
                        k.rr < rrA || k.rr >= rrB || ch + 8 > C) return -2;
                    for (int e = 0; e < 8; ++e) {
                        float m = -INFINITY;
                        for (int q = 0; q < (GA::POOLIN ? 4 : 1); ++q) {
                            const int pix = GA::cur_src_pixel(k, q);
                            if (pix < 0 || pix >= Gimg * GA::IH * GA::IW) return -2;
                            m = std::fmax(m, src[(size_t)pix * C + ch + e]);
                        }
                        x[e] = m;
                    }
                    if (GA::POOLIN && GA::cur_own(k, sbA, RA) && out_pool)       // (the halo rows too: the neighbours write the same values)
                        for (int e = 0; e < 8; ++e) out_pool[(size_t)k.spix * GA::C0 + ch0 + e] = x[e];
                    const int lo = PR::WB + k.lds;
                    if (lo < IMGA || lo + GA::ILB > MID) return -3;
                    uint8_t* dst = lds.data() + lo;
                    for (int e = 0; e < GA::ILB; ++e) wrote[lo + e] += 1;
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
            for (int q = 0; q < GA::n_buf_pixels(RA); ++q)
                if (GA::pad_pixel(q, sbA, Gimg)) {
                    std::memset(lds.data() + IMGA + (size_t)q * GA::PS, 0, GA::IPP * GA::ILB);
                    for (int e = 0; e < GA::IPP * GA::ILB; ++e) wrote[IMGA + (size_t)q * GA::PS + e] += 1;
                }
            for (int q = 0; q < GA::n_buf_pixels(RA); ++q)
                for (int e = 0; e < GA::IPP * GA::ILB; ++e)
                    if (wrote[IMGA + (size_t)q * GA::PS + e] != 1) return -6;
            // ---- B's zero rows / columns (C32Stage<GB>::zeros with the image at MID) ----
            for (int q = 0; q < GB::n_buf_pixels(R); ++q)
                if (GB::pad_pixel(q, sb, Gimg)) {
                    std::memset(lds.data() + MID + (size_t)q * GB::PS, 0, GB::IPP * GB::ILB);
                    for (int e = 0; e < GB::IPP * GB::ILB; ++e) wrote[MID + (size_t)q * GB::PS + e] += 1;
                }
            // ---- tiles of A: every valid pixel -> B's image; the member's own rows -> memory ----
            const int NTA = GA::n_tiles(RA);
            ntiles_total += NTA;
            for (int tt = 0; tt < NTA; ++tt)
                for (int sl = 0; sl < GA::SPM; ++sl) {
                    std::vector<double> acc(64 * 16);
                    const int cs = sl % GA::CS;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 16; ++r) acc[lane * 16 + r] = biasA[32 * cs + 16 * (lane >> 5) + r];
                    if (tile_acc<GA>(lds, 0, PR::WB, MID, tt, sl, acc.data())) return -4;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = lane & 31, hq = lane >> 5;
                        const typename GA::Out o = GA::out_pixel(tt, n, sbA, RA);
                        if (!o.valid) continue;
                        if (o.g < 0 || o.g >= Gimg || o.y >= GA::H || o.x < 0) return -5;
                        const bool own = o.orow >= 1 && o.orow <= R;
                        const size_t pix = (size_t)GA::out_index(o.g, o.y, o.x, 0);
                        for (int k8 = 0; k8 < 2; ++k8) {
                            const int moff = MID + PR::mid_off(o.orow, o.x, 4 * cs + 2 * hq + k8);
                            if (moff < MID || moff + GB::ILB > PR::lds_bytes(R)) return -3;
                            for (int e = 0; e < 8; ++e) {
                                const int r = 8 * k8 + e;
                                float v = (float)acc[lane * 16 + r];
                                if (RELU_A) v = v > 0.f ? v : 0.f;
                                if (MODE == C32_NATIVE) v = (float)(half_t)v;
                                if (MODE == C32_SPLIT) {
                                    const half_t h = (half_t)v;
                                    const half_t l = (half_t)(v - (float)h);
                                    std::memcpy(lds.data() + moff + 2 * e, &h, 2);
                                    std::memcpy(lds.data() + moff + 16 + 2 * e, &l, 2);
                                } else {
                                    store16(lds.data() + moff + 2 * e, v, MODE);
                                }
                                if (own) {
                                    const size_t idx = pix * GA::COUT + 32 * cs + 16 * hq + r;
                                    outA[idx] = v;
                                    writtenA[idx] += 1;
                                }
                            }
                            for (int e = 0; e < GB::ILB; ++e) wrote[moff + e] += 1;
                        }
                    }
                }
            for (int q = 0; q < GB::n_buf_pixels(R); ++q)        // every channel byte of every pixel of B's image exactly once
                for (int e = 0; e < GB::IPP * GB::ILB; ++e)
                    if (wrote[MID + (size_t)q * GB::PS + e] != 1) return -9;
            // ---- tiles of B ----
            const int NTB_ = GB::n_tiles(R);
            ntiles_total += NTB_;
            for (int tt = 0; tt < NTB_; ++tt)
                for (int sl = 0; sl < GB::SPM; ++sl) {
                    std::vector<double> acc(64 * 16);
                    const int cs = sl % GB::CS;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 16; ++r) acc[lane * 16 + r] = biasB[32 * cs + 16 * (lane >> 5) + r];
                    if (tile_acc<GB>(lds, PR::WA, PR::imgA_bytes(R), C32_LDS, tt, sl, acc.data())) return -4;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = lane & 31, hq = lane >> 5;
                        const typename GB::Out o = GB::out_pixel(tt, n, sb, R);
                        if (!o.valid) continue;
                        if (o.g < 0 || o.g >= Gimg || o.y >= GB::H || o.x < 0) return -5;
                        const size_t pix = (size_t)GB::out_index(o.g, o.y, o.x, 0);
                        for (int r = 0; r < 16; ++r) {
                            float v = (float)acc[lane * 16 + r];
                            if (RELU_B) v = v > 0.f ? v : 0.f;
                            if (MODE == C32_NATIVE) v = (float)(half_t)v;
                            const size_t idx = pix * GB::COUT + 32 * cs + 16 * hq + r;
                            outB[idx] = v;
                            writtenB[idx] += 1;
                        }
                    }
                }
        }
    }
    if (stats) { stats[0] = max_lds; stats[1] = ntiles_total; stats[2] = PR::RBMAX; stats[3] = GB::PS; }
    return 0;
}
template <class GA, class GB>
constexpr bool pair_ok() {                                       // = c32_pair_ok of giga_conv32.h
    if constexpr (GA::KIND == CONV3 && GB::KIND == CONV3 && GA::H == GB::H && GA::COUT == GB::CIN && GB::C1 == 0 && !GB::POOLIN &&
                  GA::SGM == 1 && GB::SGM == 1 && GA::KP == 1 && GB::KP == 1)
        return (C32_LDS - GA::WBYTES - GB::WBYTES - (4 * GA::ROWB + 2 * GB::ROWB + C32_TAIL * (GA::PS + GB::PS))) / (GA::ROWB + GB::ROWB) >= 6;
    else
        return false;
}
template <int MODE, int LA, int LB>
static int emu_pair_layers(const uint8_t* blob, int Gimg, const float* in0, const float* in1, float* outA, float* out_pool, int* writtenA,
                           float* outB, int* writtenB, int* stats) {
    using GA = typename U32Layer<MODE, LA>::G;
    using GB = typename U32Layer<MODE, LB>::G;
    if constexpr (pair_ok<GA, GB>()) {
        const PackOff ko = pack_offsets();
        auto W = [&](int l) { return blob + (MODE == C32_SPLIT ? ko.conv[l].c32s : MODE == C32_BF16 ? ko.conv[l].c32b : ko.conv[l].c32h); };
        auto Bv = [&](int l) { return reinterpret_cast<const float*>(blob + ko.conv[l].bias); };
        return emu_pair<GA, GB, true, true>(W(LA), Bv(LA), W(LB), Bv(LB), Gimg, in0, in1, outA, out_pool, writtenA, outB, writtenB, stats);
    } else {
        return 1;                                                // this mode runs the two layers separately
    }
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

// the fused pair (layer_a, layer_a + 1) of `mode`: 0 ok, 1 = the mode does not fuse this pair, < 0 = an emulation check failed
int conv32_emu_pair(int layer_a, int mode, const uint8_t* blob, int Gimg, const float* in0, const float* in1, float* outA, float* out_pool,
                    int* writtenA, float* outB, int* writtenB, int* stats) {
#define P(la, lb)                                                                                                                    \
    if (layer_a == la) {                                                                                                             \
        if (mode == C32_NATIVE) return emu_pair_layers<C32_NATIVE, la, lb>(blob, Gimg, in0, in1, outA, out_pool, writtenA, outB, writtenB, stats); \
        if (mode == C32_SPLIT) return emu_pair_layers<C32_SPLIT, la, lb>(blob, Gimg, in0, in1, outA, out_pool, writtenA, outB, writtenB, stats);   \
        if (mode == C32_BF16) return emu_pair_layers<C32_BF16, la, lb>(blob, Gimg, in0, in1, outA, out_pool, writtenA, outB, writtenB, stats);     \
    }
    P(0, 1) P(2, 3) P(10, 11)
#undef P
    return -11;
}

// the K-slot table of the f16-class conv_in (giga_layout.h: ci16_tap / ci16_read) and its LDS strides
void ci16_table(int* tap, int* read, int* strides) {
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 8; ++e) { tap[8 * g + e] = ci16_tap(g, e); read[8 * g + e] = ci16_read(g, e); }
    strides[0] = CI16_RS; strides[1] = CI16_SLAB;
}

}  // extern "C"
