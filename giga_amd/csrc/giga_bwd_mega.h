// Arguments of the persistent data-gradient kernel (giga_encoder.hip: unet_dgrad_mega_kernel), shared with its caller
// (giga_encoder_bwd.hip).
#pragma once
#include "giga_conv16.h"

namespace giga {

struct BwdPool {               // dS = S > 0 ? dcat[..., coff : coff + C] + unpool(dQ) : 0   (pool_bwd_add_kernel, giga_encoder_bwd.hip)
    float* dS; const float* dcat; const float* dQ; const float* S; const float* Q;
    int cs, coff, H, W, C;
};
struct BwdMegaArgs {
    ConvArgs layer[13];        // in stage order L12 L11 L10 L9 L8 L7 L6 L5 L4 L3 L2 L1 L0 (nimg = all 3B images)
    BwdPool pool[2];           // pool1 (between L4 and L3), pool0 (between L2 and L1)
    unsigned* sync;            // MEGA_SYNC_WORDS words of scratch
};
constexpr int MEGA_SYNC_WORDS = (8 + 32) * 32;    // barrier words of the persistent U-Net kernels: 8 per-XCD + 32 per-group counters, 128 B apart
int launch_unet_dgrad_mega(BwdMegaArgs m, bool bf16, hipStream_t s);

}  // namespace giga
