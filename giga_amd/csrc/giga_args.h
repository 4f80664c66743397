// Kernel-argument structs shared by the launchers (giga_decoder.hip) and the C ABI (giga_capi.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include "giga_layout.h"

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
    int heads_per_wg;        // fp32 kernel: heads handled by one workgroup (blockIdx.y selects the group)
    int post;                // 1: sigmoid(qual), normalize(rot)  (models/__init__.py:120-122)
    const float* lin;        // lattice mode: the R lattice coordinates (detection_implicit.py:28-31)
    int R;                   // lattice mode: points per axis; planes = lattice-resampled planes [3][B][R][R][32]
    float invN;              // 1 / N
    unsigned mR, mR2;        // ceil(2^32 / R), ceil(2^32 / R^2)   (lattice mode)
    int lat_parts;           // decoder_lat_kernel: a slab (scene, ix) is handed out in this many parts (tile ranges): small batches
};

// Byte offsets of the encoder's activations inside its caller-allocated workspace (giga_encoder.hip::enc_workspace).
struct EncWs {
    size_t P0, A0, S0, Q0, A1, S1, Q1, A2, S2, U0, A3, A4, U1, A5, A6, YZ, XZ, total;   // XZ unused (kept for the ABI)
    size_t SYNC;               // barrier counter + timeout flag of the persistent U-Net kernel
    size_t MASK;               // conv_in ReLU mask of a training forward (GIGA_CONVIN_MASK): [B][8][40][64 lanes] x 16 bytes
};

// gradient workspace of the encoder backward (giga_encoder_bwd.hip carves it, giga_capi.hip sizes the caller's buffer with it): ONE
// definition for both translation units
struct BwdWs { size_t gA6, gA5, gC1, gA4, gA3, gC0, gS2, gA2, gQ1, gS1, gA1, gQ0, gS0, gA0, gP0, WG, WG3[12], CINP, total; };   // (+ SYNC behind `total`)
BwdWs enc_bwd_workspace(int B);

}  // namespace giga
