/* giga_hip.h -- C ABI of the MI355X-native GIGA hot path (libgiga_hip.so).
 *
 * The reference (UT-Austin-RPL/GIGA) has no FFI layer: the boundary it exposes for this path is the
 * nn.Module surface consumed by scripts/train_giga.py:204 and src/vgn/detection_implicit.py:107.
 * `giga_amd/` keeps that Python surface (same class names, kwargs, tensor layouts and state-dict
 * keys) and calls the functions below underneath; each entry point cites the reference code it
 * replaces (paths relative to /root/reference/src/vgn).  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every device buffer is owned by the caller (PyTorch);
 *     the library never allocates device memory and is re-entrant per (stream, workspace).  Its only
 *     process-wide mutable state is bookkeeping, never read by the arithmetic: the launch counter
 *     (giga_launch_count), the per-kernel record of raised dynamic-LDS limits, and per device the
 *     completion events of the streams' last persistent U-Net launches (see GIGA_PERSIST_UNET);
 *   - every launch is asynchronous on the caller's HIP stream `stream` (a hipStream_t);
 *   - return value 0 = success, negative = error (giga_strerror); nothing throws across the ABI;
 *   - at most GIGA_MAX_SCENES scenes per encoder / training call (the convolution kernels address activations with
 *     24-bit pixel indices and 32-bit byte offsets); larger batches return -7, split them;
 *   - `precision`: 0 = exact fp32 (v_mfma_f32_32x32x2_f32, bitwise fp32 fma chains),
 *                  1 = f16 operands / fp32 accumulate (v_mfma_f32_32x32x16_f16);
 *                  2 = f16x3 split operands / fp32 accumulate: every operand is the pair hi = f16(v), lo = f16(v - hi)
 *                      and every product W_lo*x_hi + W_hi*x_lo + W_hi*x_hi on the f16 MFMA (~22-bit operands, results
 *                      within 1e-5 of the fp32 path).  Decoder entry points only change arithmetic; planes are fp32
 *                      (as for precision 0); the encoder runs its convolutions (conv_in and the U-Net) in the same split
 *                      arithmetic;
 *                  3 = bf16 operands / fp32 accumulate, fp32 activations and planes in memory: the forward of the bf16 training
 *                      step.  Encoder entry points: the U-Net convolutions (fp32 conv_in; see GIGA_BF16_CONVS); giga_decoder_forward:
 *                      every linear layer of the heads (see GIGA_BF16_DECODER; not with GIGA_FOLD_FINAL, not the lattice entry point);
 *   - "NHWC planes": one buffer [3 (xz,xy,yz)][B][40 (H)][40 (W)][32 (C)] of float (precision 0 and 2) or
 *     _Float16 (precision 1).  H/W follow the reference's plane indexing (ConvONets/common.py:246-251,
 *     303-318): xz -> (H=z, W=x), xy -> (H=y, W=x), yz -> (H=z, W=y).
 */
#ifndef GIGA_HIP_H_
#define GIGA_HIP_H_

#define GIGA_MAX_SCENES 3072

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIGA_ABI_VERSION 3   /* 3: the forward blob grew (Winograd-domain images of the fp32 3x3 layers, round 6); 2: bf16 decoder images: repack */

#define GIGA_HEAD_QUAL 1   /* decoder_qual  (out_dim 1, sigmoid epilogue)     */
#define GIGA_HEAD_ROT 2    /* decoder_rot   (out_dim 4, L2-normalise epilogue) */
#define GIGA_HEAD_WIDTH 4  /* decoder_width (out_dim 1)                        */
#define GIGA_HEAD_TSDF 8   /* decoder_tsdf  (out_dim 1, raw logits)            */

int giga_abi_version(void);
const char* giga_strerror(int code);

/* Number of fp32 parameters expected for a head set (bitmask of GIGA_HEAD_*): the reference
 * state-dict flattened in its own key order (networks.py:21-35 load_network / state-dict keys of
 * conv_onet/models/__init__.py:15-40).  giga: 581863 for all four heads. */
size_t giga_param_count(int head_present);

/* Size of the packed-weight blob produced by giga_pack_weights. */
size_t giga_packed_bytes(void);

/* HOST function: repack the flat fp32 parameters (host pointer) into MFMA operand fragments
 * (host pointer, giga_packed_bytes() bytes).  The caller uploads the blob to the device.
 * Replaces the weight ownership of nn.Module.load_state_dict (networks.py:33-34). */
int giga_pack_weights(const float* params_host, size_t n_params, int head_present,
                      void* packed_host, size_t packed_bytes);

/* The blob layout belongs to the ABI version: new images are appended and slot orders change between versions, and the compute entry
 * points take no blob size -- a blob packed by another version would be read at the wrong offsets, silently.  Blobs are not
 * portable across versions: REPACK from the parameters after an upgrade.  Both packers stamp the last 256 bytes of their blob (magic,
 * ABI version, size); giga_packed_check validates a HOST copy (`backward` != 0: a giga_pack_bwd_weights blob) before it is uploaded:
 * 0, or -8 for a foreign / truncated / stale blob.  (A blob rebuilt on the device by giga_repack_device carries no stamp.) */
int giga_packed_check(const void* packed_host, size_t bytes, int backward);

/* Training support (scripts/train_giga.py:198-211: weights change every optimizer step): the fp32
 * words of the blob are pure gathers of single parameters.  giga_pack_map fills a HOST int32 map with one
 * entry per 4-byte blob word (>= 0 parameter index, -1 constant zero, -2 not an fp32 word);
 * giga_repack_device rewrites the fp32 words of a DEVICE blob from a DEVICE flat parameter buffer
 * (asynchronous on `stream`; f16 fragments are left untouched). nwords = giga_packed_bytes() / 4. */
int giga_pack_map(int head_present, int32_t* map_host, size_t nwords);
int giga_repack_device(const float* params_dev, const int32_t* map_dev, void* packed_dev, size_t nwords,
                       void* stream);
/* the forward and the backward blob in ONE launch (the training step rebuilds both every step) */
int giga_repack_device2(const float* params_dev, const int32_t* map_fwd_dev, void* packed_dev, size_t nwords_fwd,
                        const int32_t* map_bwd_dev, void* bwd_packed_dev, size_t nwords_bwd, void* stream);

/* Scratch bytes giga_encoder_forward needs for a batch of B scenes. */
size_t giga_encoder_workspace_bytes(int B, int precision);

/* Introspection for tests: byte offsets of the 17 intermediate activations inside the encoder
 * workspace, in order P0 (projected planes, U-Net input) A0 S0 Q0 A1 S1 Q1 A2 S2 U0 A3 A4 U1 A5 A6
 * YZ XZ (see giga_encoder.hip).  All NHWC [3B][H][W][C] in the precision's element type
 * (YZ/XZ: fp32 partial sums).  offsets must have room for 17 entries. */
int giga_encoder_workspace_layout(int B, int precision, size_t* offsets);

/* LocalVoxelEncoder.forward (encoder/voxels.py:89-121) + UNet.forward (encoder/unet.py:225-239):
 * tsdf [B][40][40][40] fp32  ->  feature planes.
 *   planes_nhwc : NHWC planes in the precision's element type (consumed by giga_decoder_forward)
 *   planes_nchw : optional (may be NULL) fp32 [3][B][32][40][40], i.e. the reference's
 *                 {'xz','xy','yz'} dict of (B,32,40,40) tensors stacked. */
int giga_encoder_forward(const float* tsdf, const void* packed, void* planes_nhwc, float* planes_nchw,
                         int B, int precision, void* workspace, size_t workspace_bytes, void* stream);

/* Repack reference-layout planes (three fp32 (B,32,40,40) tensors) into NHWC planes; used when
 * LocalDecoder.forward(p, c_plane) (conv_onet/models/decoder.py:133) is handed foreign planes. */
int giga_planes_pack(const float* xz, const float* xy, const float* yz, void* planes_nhwc, int B,
                     int precision, void* stream);
/* Inverse: NHWC planes -> fp32 [3][B][32][40][40]. */
int giga_planes_unpack(const void* planes_nhwc, float* planes_nchw, int B, int precision, void* stream);

/* Fused LocalDecoder.forward for every head in `head_mask` (decoder.py:133-176; ResnetBlockFC
 * layers.py:39-47) over p [B][N][3] fp32 in [-0.5,0.5]^3 (any value is clamped like
 * common.py:238-261).  Point g uses the planes of scene g / N.
 *   qual [B*N], rot [B*N][4], width [B*N], occ [B*N]; a pointer may be NULL iff its head is not
 *   in head_mask.
 *   post != 0 applies the epilogues of ConvolutionalOccupancyNetwork.decode
 *   (conv_onet/models/__init__.py:111-124): sigmoid(qual), F.normalize(rot, dim=2). */
int giga_decoder_forward(const void* planes_nhwc, const float* p, const void* packed, int head_mask,
                         float* qual, float* rot, float* width, float* occ, int B, int N,
                         int precision, int post, void* stream);

/* GIGA_FOLD_FINAL, OR-ed into `precision` of giga_encoder_forward* AND of the giga_decoder_forward* call that consumes its
 * planes: the encoder stops before its last layer, conv_final (a 1x1 convolution without activation, encoder/unet.py:238),
 * and the decoder uses head images in which that layer is folded into fc_c -- bilinear sampling (decoder.py:117-122)
 * commutes with a per-pixel linear map, so fc_c(sample(Wf x + bf)) = (Wc blockdiag(Wf)) sample(x) + (bc + Wc bf).  Same
 * outputs (fp32 rounding-level differences), one layer less.  planes_nchw must be NULL with this flag: the planes it
 * produces are NOT LocalVoxelEncoder's return value and are only meaningful to a decoder call carrying the flag. */
#define GIGA_FOLD_FINAL 16
/* How the twelve (thirteen) U-Net layers are launched.  DEFAULT: ONE persistent launch in which groups of 8 workgroups, each
 * group inside one XCD, walk their share of the 3 * B plane images through all layers with a barrier among those 8 only
 * (csrc/giga_encoder.hip::unet_mega_kernel) -- for every batch size in precisions 1 and 2, from 8 scenes up in precisions 0
 * and 3.  Same results as one launch per layer: bit for bit in the f16-class modes, to fp32 rounding (<= 2e-6 relative) in
 * precision 0, where the summation order of a layer's ragged last round follows the work distribution.  A layer boundary then
 * costs a 1-us barrier instead of a launch: the whole encoder takes 75 instead of 94 us for one scene in precision 1 (the
 * planner of detection_implicit.py:99-113 runs one scene at a time), 130 instead of 150 us at 32 scenes, 361 instead of 378 in
 * precision 0.  The workgroups of a group find each other by ticket among the workgroups already resident on their XCD, so
 * such launches may be in flight on several streams of one device without waiting on each other's CUs -- UP TO FOUR of them: a
 * launch can hold one unfilled group (<= 7 workgroups) per XCD, an XCD has 32 slots for these workgroups, and 5 x 7 > 32 could
 * park every slot in groups that never fill.  The library enforces the bound inside a process: per device it tracks the
 * completion event of every stream's last persistent launch, and a call on a stream that would be the fifth with an unfinished one
 * takes one launch per layer instead (same results; launches queued on one stream run one after the other and count once).  It cannot see OTHER PROCESSES that share the device, nor replays of captured hipGraphs on several streams at
 * once: keep those to four concurrent encoder calls, or pass GIGA_LAYERWISE_UNET.  The kernel also assumes that every XCD
 * receives an eighth of the grid (no CU mask on the stream, a whole MI355X: 256 CUs -- checked -- in one partition).  A barrier
 * that is not released within 20 s of wall clock (s_memrealtime) traps instead of returning stale data or hanging the device.
 *   GIGA_LAYERWISE_UNET, OR-ed into `precision` of giga_encoder_forward*: one launch per layer (A/B comparisons; the environment
 *   variable GIGA_UNET_PERSIST=0 does the same for a whole process).
 *   GIGA_PERSIST_UNET: the persistent launch also where the default keeps per-layer launches (precisions 0 / 3 below 8 scenes). */
#define GIGA_PERSIST_UNET 32
#define GIGA_LAYERWISE_UNET 64
/* Which convolution kernels run the U-Net of the f16-class precisions (1 and 2):
 *   conv32 (csrc/giga_conv32.h; round 4; the default up to 16 scenes): v_mfma_f32_32x32x16, a member of a group owns a band of ROWS of the
 *     group's images -- staged once per layer into LDS -- and one LDS-DMA copy of its weights; register tiles fed by conflict-free
 *     ds_read_b128 at `base + immediate`, no VALU in the MFMA loop; at up to two images per group the same-resolution layer
 *     pairs (0,1), (2,3), (10,11) of plain f16 run without the group barrier between them (the second layer reads the first one's
 *     output from LDS; one halo row recomputed per side; bit-identical to the unfused form).
 *   conv16 (csrc/giga_conv16.h): 16x16x32, wave-private haloed patches; the only kernels of precisions 0 and 3.
 * Same arithmetic per layer (f16 / f16x3 operands, fp32 accumulation, outputs within one rounding of each other), same launch
 * forms, same workspace.  Default: conv32 up to 16 scenes (48 images), conv16 beyond.  Measured: conv32's
 * encoder is 3-9 % faster up to 32 scenes, but with every CU busy (from ~24 scenes) the chip holds a 3-5 % lower shader clock while
 * and after the conv32 launch runs, which costs sustained large-batch steps more than the kernel gains.  A scene's low-order bits
 * therefore depend on whether its batch has more than 16 scenes -- force one kernel where that matters:
 *   GIGA_CONV32_UNET / GIGA_CONV16_UNET, OR-ed into `precision` (1 or 2) of giga_encoder_forward*: force the one or the other for
 *   this call; the environment variable GIGA_CONV32=1 / 0 does the same for a whole process (the flag of a call wins). */
#define GIGA_CONV32_UNET 128
#define GIGA_CONV16_UNET 256
/* GIGA_CONVIN_MASK, OR-ed into `precision` (0 or 3) of giga_encoder_forward*: a TRAINING forward -- conv_in also stores the sign bits of
 * its pre-activations (10 MB at 32 scenes, in the encoder workspace), and giga_backward called with GIGA_CONVIN_MASK_BWD on that workspace
 * takes the ReLU mask of conv_in from them instead of recomputing the convolution (8 instead of 15 MFMAs per unit). */
#define GIGA_CONVIN_MASK 512
/* The stride-1 3x3 layers of the U-Net in precision 0 (exact fp32; encoder/unet.py:14-23) run as Winograd F(2x2, 3x3) on the fp32
 * MFMA by default (csrc/giga_wino.h: 16 instead of 36 multiplies per 2x2 outputs; transforms in registers, Winograd-domain weights
 * from the packed blob).  fp32 Winograd is not the direct form's bitwise fma chain: planes differ from it by a few 1e-6 relative
 * (the suite holds both to 1e-4 of the oracle).  GIGA_DIRECT_CONV, OR-ed into `precision` (0) of giga_encoder_forward*, keeps the
 * direct convolutions for this call; the environment variable GIGA_WINOGRAD=0 does so for a whole process (a layer bit mask
 * otherwise).  A blob rebuilt by giga_repack_device (the training path) needs giga_derive_winograd before such a forward:
 * giga_repack_device gathers parameters, and the Winograd image is not a gather. */
#define GIGA_DIRECT_CONV 1024

/* Inference fast path for the FIXED QUERY LATTICE of VGNImplicit (detection_implicit.py:28-31,107):
 * the R^3 points meshgrid(lin, lin, lin, 'ij') with z fastest, shared by all B scenes.  Each plane is
 * sampled at only R*R distinct positions, so the planes are first resampled at the lattice coordinates
 * (same arithmetic as sample_plane_feature, decoder.py:117-122) into `workspace`
 * (giga_lattice_workspace_bytes) and the fused decoder then reads three pixels per point instead of
 * twelve bilinear taps.  lin: device pointer to the R lattice coordinates (R <= 64).
 * Outputs as giga_decoder_forward with N = R^3; ev_start/ev_stop (may be NULL) bracket the decoder launch. */
/* GIGA_PLANES_FP32, OR-ed into `precision` 1 of giga_decoder_forward_lattice: planes_nhwc are FP32 planes (an encoder at precision 0 or
 * 2); the resampling reads them and writes the f16 lattice planes the plain-f16 decoder consumes.  The throughput decoder under an
 * fp32-grade encoder: most of plain f16's error against the fp32 reference comes from its ENCODER (tests/test_f16_error_budget.py). */
#define GIGA_PLANES_FP32 1024
size_t giga_lattice_workspace_bytes(int B, int R, int precision);
int giga_decoder_forward_lattice(const void* planes_nhwc, const float* lin, const void* packed, int head_mask,
                                 float* qual, float* rot, float* width, float* occ, int B, int R, int precision,
                                 int post, void* workspace, size_t workspace_bytes, void* stream,
                                 void* ev_start, void* ev_stop);

/* ---- training path, fp32 (scripts/train_giga.py:198-211: forward, loss.backward(), optimizer.step()) -----
 * The forward is giga_encoder_forward + giga_decoder_forward at precision 0 with the encoder workspace
 * kept alive (it holds every U-Net activation).  giga_backward computes the gradient of a scalar loss with
 * respect to EVERY parameter, given the gradients of the four head outputs:
 *   outs / douts : arrays of 4 device pointers (qual [B*N], rot [B*N][4], width [B*N], occ [B*M]); entries of
 *                  absent heads are ignored, a NULL entry of a head that runs (grasp heads: N > 0; occupancy head:
 *                  M > 0 and p_tsdf given) is GIGA -6.  outs are the forward results (post sigmoid / normalize).
 *                  Every argument is validated before anything is enqueued on the stream.
 *   grads        : flat fp32 buffer in reference state-dict order (giga_param_count), overwritten.
 *   head_present : head bits, optionally | GIGA_DETACH_OCC: the occupancy head then reads detached planes, i.e. its
 *                  loss does not reach the encoder (detach_tsdf of `giga_detach`, models/__init__.py:61-63,
 *                  networks.py:143-169); the three grasp heads are unaffected.
 * It replaces autograd through conv_onet/models/__init__.py:42-67 (decoder.py:117-176, encoder/voxels.py:89-121,
 * encoder/unet.py:225-239).  Data-gradient convolutions and the decoder's gradient chain run on MFMA with the
 * transposed weights of the BACKWARD blob (giga_bwd_packed_bytes / giga_pack_bwd_weights, host; or
 * giga_pack_bwd_map + giga_repack_device on the device every step).  Weight gradients are reduced with fp32
 * atomics (run-to-run differences at rounding level, as in PyTorch's own GPU backward). */
#define GIGA_DETACH_OCC 16
/* GIGA_BF16_CONVS, OR-ed into giga_backward's head_present: the thirteen data-gradient convolutions of the U-Net (from the
 * backward blob's bf16 images) and the weight gradients of its eleven 3x3 layers run on bf16 MFMA (operands rounded to bf16,
 * fp32 accumulate, fp32 gradients in memory); pairs with an encoder forward at precision 3.  ConvTranspose / 1x1 weight
 * gradients, the decoder and conv_in stay fp32.  BASELINE config c5 ("bf16"). */
#define GIGA_BF16_CONVS 32
/* GIGA_BF16_DECODER, OR-ed into giga_backward's head_present: the decoder heads' backward runs as ONE fused bf16 kernel per call
 * (csrc/giga_decoder_train16.hip): the forward chain recomputed on bf16 MFMA, the gradient chain with the transposed bf16 matrices,
 * and the weight gradients dW = dY^T X in the same kernel (tiles handed over through LDS and read back transposed with
 * ds_read_b64_tr_b16; gradient tiles resident in registers across the workgroup's points) -- no [P][32] row arrays, no separate
 * weight-gradient launch; one reduce launch for every head of the step.  Pairs with giga_decoder_forward at precision 3 (the same
 * bf16 arithmetic; the backward recomputes it bit for bit).  Operands (features, activations, gradients, weights) are rounded to
 * bf16 once, accumulation and the residual stream are fp32, fc_p / the biases are carried as hi + lo bf16 pairs. */
#define GIGA_BF16_DECODER 64
/* GIGA_CONVIN_MASK_BWD, OR-ed into giga_backward's head_present: enc_workspace_fwd comes from a forward with GIGA_CONVIN_MASK. */
#define GIGA_CONVIN_MASK_BWD 128
/* bf16 images of the convolution fragments, derived ON THE DEVICE from the fp32 fragments of the same blob(s) after
 * giga_repack_device (either pointer may be NULL).  giga_pack_weights / giga_pack_bwd_weights fill them on the host too. */
int giga_derive_bf16_fragments(void* packed_dev, void* bwd_packed_dev, void* stream);
/* The Winograd-domain images of the fp32 3x3 layers (csrc/giga_wino.h), derived ON THE DEVICE from the fp32 fragments of the same blob
 * after giga_repack_device: U = G g G^T accumulated in double in the host packer's order -- bit-identical to giga_pack_weights' /
 * giga_pack_bwd_weights'.  packed_dev: the forward blob (a precision-0 forward on a device-repacked blob needs it unless the call carries
 * GIGA_DIRECT_CONV); bwd_packed_dev: the backward blob (the fp32 data-gradient chain of giga_backward runs its 3x3 layers as Winograd
 * too unless GIGA_WINOGRAD_BWD=0 is set in the environment).  Either may be NULL. */
int giga_derive_winograd(void* packed_dev, void* bwd_packed_dev, void* stream);
size_t giga_bwd_packed_bytes(void);
int giga_pack_bwd_weights(const float* params_host, size_t n_params, int head_present, void* packed_host,
                          size_t packed_bytes);
int giga_pack_bwd_map(int head_present, int32_t* map_host, size_t nwords);
size_t giga_backward_workspace_bytes(int B, int N, int M, int head_present);
/* Introspection for tests: byte offsets, inside the backward workspace, of the 15 activation-gradient tensors the encoder backward
 * leaves there (fp32 NHWC [3B][H][W][C], in order gA6 gA5 gC1 gA4 gA3 gC0 gS2 gA2 gQ1 gS1 gA1 gQ0 gS0 gA0 gP0: g<X> = gradient of
 * the loss w.r.t. activation X of giga_encoder_workspace_layout; gC1 / gC0 = w.r.t. the concatenated inputs of up1.conv1 /
 * up0.conv1).  gA6 gA5 gA4 gA3 gS2 gA2 gS1 gA1 gS0 gA0 are, with the ReLU mask already applied, the dY operands of the weight
 * gradients of layers 11, 10, 8, 7, 5, 4, 3, 2, 1, 0.  offsets must have room for 15 entries. */
int giga_backward_workspace_layout(int B, size_t* offsets);
int giga_backward(const float* tsdf, const void* packed, const void* bwd_packed, const void* enc_workspace_fwd,
                  const void* planes_nhwc, const float* p, const float* p_tsdf, const float* const* outs,
                  const float* const* douts, float* grads, size_t n_params, int head_present, int B, int N, int M,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- input side: page-locking of caller-owned host memory (hipHostRegister / hipHostUnregister), so that batches which
 * reader processes assemble in a shared-memory ring are DMA-ed to the device without a staging copy
 * (giga_amd/feed.py; replaces the pageable `.to(device)` of train_giga.py:141-151). */
int giga_host_register(void* ptr, size_t bytes);
int giga_host_unregister(void* ptr);

/* ---- input side: dense export of a sparse TSDF (src/vgn/perception.py:107-115, TSDFVolume.get_grid) -----
 * voxel_index [n][3] int32 (grid_index i, j, k) and voxel_value [n] float32 (voxel.color[0]) list the observed voxels of B
 * scenes back to back; scene b owns voxels scene_offsets[b] .. scene_offsets[b+1]-1 (scene_offsets [B+1], device).
 * grid [B][R][R][R] float32 receives 0 for unobserved cells and the value of the LAST voxel of the list that targets a
 * cell (the Python loop's semantics; Open3D's lists have unique indices).  Indices outside [0, R) are ignored.
 * workspace: giga_tsdf_scatter_workspace_bytes(B, R) bytes of device scratch. */
size_t giga_tsdf_scatter_workspace_bytes(int B, int R);
int giga_tsdf_scatter(const int32_t* voxel_index, const float* voxel_value, const int32_t* scene_offsets, int B, int R,
                      int n_voxels, float* grid, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused training loss (scripts/train_giga.py:154-195: `select` + `loss_fn`, literal call shape N = 1) -----
 * Inputs are the head outputs of the model's forward for one grasp query per scene -- qual [B] (post sigmoid), rot [B][4]
 * (unit), width [B], occ_logits [B][M] (raw; the sigmoid of `select` is part of this function) -- and the labels of
 * `prepare_batch` (train_giga.py:141-151): label [B], rot_targets [B][2][4], width_target [B], occ_target [B][M].
 *   giga_train_loss          : losses [5] (device) = means over the batch of loss_qual, loss_rot, loss_width, loss_occ and
 *                              loss_all = mean(loss_qual + label (loss_rot + 0.01 loss_width) + loss_occ), the `loss_dict`
 *                              of train_giga.py:169-173; scene_losses [B][5] (device, scratch and per-scene values).
 *   giga_train_loss_backward : d loss_all / d(qual, rot, width, occ_logits), scaled by the device scalar *grad_loss.
 * Arithmetic follows ATen's fp32 formulas (BCE logs clamped at -100, BCE gradient (p-y)/max(p(1-p),1e-12)); the mean
 * over scenes is summed in scene order (deterministic). */
int giga_train_loss(const float* qual, const float* rot, const float* width, const float* occ_logits, const float* label,
                    const float* rot_targets, const float* width_target, const float* occ_target, int B, int M,
                    float* losses, float* scene_losses, void* stream);
int giga_train_loss_backward(const float* qual, const float* rot, const float* width, const float* occ_logits,
                             const float* label, const float* rot_targets, const float* width_target,
                             const float* occ_target, const float* grad_loss, int B, int M, float* dqual, float* drot,
                             float* dwidth, float* docc, void* stream);

/* Adam over ONE flat fp32 buffer (the flattened parameters of the module, `net.flatten_parameters()`), in place: the update of
 * torch.optim.Adam (the reference's optimiser, scripts/train_giga.py:49; no amsgrad), formulas and order of torch/optim/adam.py:
 * m = lerp(m, g, 1 - beta1); v = beta2 v + (1 - beta2) g^2; p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps);
 * weight_decay (L2) adds weight_decay * p to g first.  `step` counts from 1.  One launch over the whole chip: torch's fused
 * multi-tensor kernel gives a single tensor one workgroup per 65 536 elements (98 us for these 581 863 parameters, this: ~5 us).
 * All four buffers must be 16-byte aligned (-1 otherwise). */
int giga_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, int step, void* stream);

/* ---- grasp post-processing (src/vgn/detection_implicit.py:115-143 process, :87-97 bound, :146-174 select) -----
 * Replaces the host-side scipy stage that follows `predict` in VGNImplicit.__call__ (detection_implicit.py:55-58)
 * for B scenes at once; every volume is [B][R][R][R] float32 (rot: [B][R^3][4]) on the device.
 *   process : gaussian_filter(qual, sigma, mode="nearest"); qual = 0 where the voxel is not in the 2-iteration
 *             masked binary dilation of (tsdf > out_th) with mask ~(1e-3 < tsdf < out_th), or where
 *             width < min_width or width > max_width                      (detection_implicit.py:126-141)
 *   bound   : qual = 0 for x < lim_x, x >= R-lim_x, y < lim_y, y >= R-lim_y, z < lim_z; the caller computes
 *             lim = int(limit / voxel_size) as detection_implicit.py:89-91 does
 *   select  : qual < low_th -> 0; if force_detection and no voxel >= threshold the scene is "best only" and
 *             the threshold is skipped, else qual < threshold -> 0; NMS with maximum_filter(size =
 *             max_filter_size, mode reflect); survivors are appended (unordered) to the candidate lists.
 * Outputs: qual_out [B][R^3] = the volume after process+bound (what the reference hands to select and to its
 * visualiser); counters [B][2] = {#voxels >= threshold (0 => best-only when force_detection), #candidates};
 * cand_index/score/width [B][cap], cand_rot [B][cap][4] hold the first min(#candidates, cap) survivors: flat
 * voxel index (x*R+y)*R+z, score, quaternion and width.  The caller sorts by descending score
 * (detection_implicit.py:166-171) and keeps one entry for a best-only scene. */
typedef struct GigaGraspParams {
    double gaussian_sigma;     /* 1.0 */
    float min_width;           /* 0.033 */
    float max_width;           /* 0.233 */
    float out_th;              /* VGNImplicit(out_th=0.5) */
    float low_th;              /* LOW_TH = 0.5, detection_implicit.py:15 */
    float threshold;           /* VGNImplicit(qual_th=0.9) */
    int lim_x, lim_y, lim_z;   /* int(0.02/voxel_size), int(0.02/voxel_size), int(0.055/voxel_size) */
    int max_filter_size;       /* 4 (8 when visualising) */
    int force_detection;
} GigaGraspParams;
size_t giga_grasp_workspace_bytes(int B, int R);
int giga_grasp_select(const float* tsdf, const float* qual, const float* rot, const float* width, int B, int R,
                      const GigaGraspParams* params, float* qual_out, int* counters, int cap, int* cand_index,
                      float* cand_score, float* cand_rot, float* cand_width, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Number of kernel launches the library has enqueued in this process so far (all streams): the difference around a call is
 * that call's launch count.  Diagnostic only. */
unsigned long long giga_launch_count(void);
/* Measurement hook for ANY launch of the library (bench.py: the kernels of the training step): the launch whose ordinal -- the value
 * giga_launch_count() returns right after it was enqueued -- equals `ordinal` is bracketed by ev_start / ev_stop (giga_event_create)
 * on the stream it is launched on; giga_launch_probe_name() then returns the kernel expression of that launch site.  ordinal 0
 * disarms.  One probe per process; not for concurrent use from several host threads. */
int giga_launch_probe(unsigned long long ordinal, void* ev_start, void* ev_stop);
const char* giga_launch_probe_name(void);
/* The library's per-device bookkeeping -- which kernels have had their dynamic-LDS limit raised, which streams have a persistent
 * U-Net launch in flight -- outlives a hipDeviceReset(), the state it describes does not.  A host that resets a device calls this
 * before its next call into the library (no GPU work is enqueued; safe to call at any time when no call is in progress). */
void giga_forget_device_state(void);
/* How the LAST encoder call of this process ran its U-Net (diagnostic: tests pin the launch-form and kernel flags to it):
 * an OR of GIGA_PATH_PERSISTENT (one persistent launch; else one launch per layer), GIGA_PATH_CONV32 (conv32 kernels; else
 * conv16), GIGA_PATH_FUSED_PAIRS (the persistent conv32 launch ran its same-resolution layer pairs fused) and GIGA_PATH_WINOGRAD
 * (precision 0: 3x3 layers ran as Winograd F(2x2, 3x3)). */
#define GIGA_PATH_PERSISTENT 1
#define GIGA_PATH_CONV32 2
#define GIGA_PATH_FUSED_PAIRS 4
#define GIGA_PATH_WINOGRAD 8
int giga_encoder_last_path(void);
/* ---- measurement hooks (bench.py roofline): HIP events recorded on `stream` right before and
 * after ONE kernel launch, so the kernel's duration is measured live on the stream it runs on.
 * probe_stage: 0 conv_in+project, 1 plane_finalize, 2..14 = U-Net layers 0..12 (giga_layout.h kConv), 15 = the whole U-Net.
 * The decoder variant brackets the single fused decoder launch. */
void* giga_event_create(void);
void giga_event_destroy(void* ev);
int giga_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises on ev_stop */
int giga_event_record(void* ev, void* stream);   /* bench.py: an EMPTY bracket measures what the two records themselves cost */
int giga_encoder_forward_probe(const float* tsdf, const void* packed, void* planes_nhwc, float* planes_nchw,
                               int B, int precision, void* workspace, size_t workspace_bytes, void* stream,
                               int probe_stage, void* ev_start, void* ev_stop);
int giga_decoder_forward_probe(const void* planes_nhwc, const float* p, const void* packed, int head_mask,
                               float* qual, float* rot, float* width, float* occ, int B, int N,
                               int precision, int post, void* stream, void* ev_start, void* ev_stop);

#ifdef __cplusplus
}
#endif
#endif /* GIGA_HIP_H_ */
