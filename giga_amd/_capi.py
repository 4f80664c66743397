"""ctypes binding of libgiga_hip.so (C ABI in include/giga_hip.h).

The library is built in-tree (`giga_amd/lib/libgiga_hip.so`, see `giga_amd/build.py`) and is the ONLY
compute path: there is no CPU or eager-PyTorch fallback.  Loading fails loudly if it is missing.
torch must be imported first so that the HIP runtime the library binds to (SONAME libamdhip64.so.7)
is the one PyTorch already loaded -- device pointers and streams are then shared.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgiga_hip.so")

HEAD_QUAL, HEAD_ROT, HEAD_WIDTH, HEAD_TSDF = 1, 2, 4, 8
DETACH_OCC = 16          # GIGA_DETACH_OCC: flag for giga_backward's head_present
BF16_CONVS = 32          # GIGA_BF16_CONVS: dgrad convolutions on bf16 MFMA (giga_backward's head_present)
BF16_DECODER = 64        # GIGA_BF16_DECODER: the decoder backward as one fused bf16 kernel per call (giga_backward's head_present)
CONVIN_MASK = 512        # GIGA_CONVIN_MASK: training forward keeps conv_in's ReLU mask (encoder `precision` flag)
CONVIN_MASK_BWD = 128    # GIGA_CONVIN_MASK_BWD: giga_backward reads it (head_present flag)
ENC_BF16 = 3             # encoder `precision` 3: bf16 U-Net convolutions, fp32 activations in memory
DEC_BF16 = 3             # decoder `precision` 3: bf16 linear layers on fp32 planes (the forward of the bf16 training decoder)
FOLD_FINAL = 16          # GIGA_FOLD_FINAL: OR-ed into `precision` of an encoder call and of the decoder calls on its planes
PERSIST_UNET = 32        # GIGA_PERSIST_UNET: OR-ed into `precision` of an encoder call (one persistent U-Net launch)
LAYERWISE_UNET = 64      # GIGA_LAYERWISE_UNET: one launch per U-Net layer even for small batches
CONV32_UNET = 128        # GIGA_CONV32_UNET: the f16-class U-Net on the conv32 kernels (the library's default)
CONV16_UNET = 256        # GIGA_CONV16_UNET: ... on the conv16 kernels
DIRECT_CONV = 1024       # GIGA_DIRECT_CONV: precision 0 keeps the direct 3x3 convolutions (default: Winograd F(2x2, 3x3), csrc/giga_wino.h)
PATH_PERSISTENT, PATH_CONV32, PATH_FUSED_PAIRS, PATH_WINOGRAD = 1, 2, 4, 8          # giga_encoder_last_path()
MAX_SCENES = 3072        # GIGA_MAX_SCENES: scenes per encoder / training call (error -7 beyond)
HEAD_BITS = {"decoder_qual": 1, "decoder_rot": 2, "decoder_width": 4, "decoder_tsdf": 8}
PLANES_FP32 = 1024       # GIGA_PLANES_FP32: fp32 planes into the plain-f16 lattice decoder
# include/giga_hip.h `precision` (3: the training forward's arithmetic).  "fp16x3+fp16" (4, Python-side only): the f16x3 encoder
# (fp32-grade planes) under the plain-f16 LATTICE decoder; its generic-query decoder is the f16x3 one
PRECISION = {"fp32": 0, "fp16": 1, "fp16x3": 2, "bf16": 3, "fp16x3+fp16": 4}
ENCODER_PRECISION = {0: 0, 1: 1, 2: 2, 3: 3, 4: 2}
PLANE_DTYPE = {0: torch.float32, 1: torch.float16, 2: torch.float32, 3: torch.float32, 4: torch.float32}   # element type of the NHWC planes
DECODER_PRECISION = {0: 0, 1: 1, 2: 2, 3: 0, 4: 2}   # "bf16" = bf16 U-Net convolutions; the decoders run their fp32 kernels
LATTICE_PRECISION = {0: 0, 1: 1, 2: 2, 3: 0, 4: 1 | PLANES_FP32}

_lib = None

_SIGNATURES = {
    "giga_abi_version": (ctypes.c_int, []),
    "giga_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "giga_param_count": (ctypes.c_size_t, [ctypes.c_int]),
    "giga_packed_bytes": (ctypes.c_size_t, []),
    "giga_pack_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_size_t]),
    "giga_packed_check": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]),
    "giga_pack_map": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]),
    "giga_repack_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p]),
    "giga_repack_device2": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "giga_encoder_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "giga_encoder_workspace_layout": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "giga_encoder_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "giga_planes_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "giga_planes_unpack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p]),
    "giga_decoder_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "giga_derive_bf16_fragments": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "giga_derive_winograd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "giga_bwd_packed_bytes": (ctypes.c_size_t, []),
    "giga_pack_bwd_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_size_t]),
    "giga_pack_bwd_map": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]),
    "giga_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "giga_backward_workspace_layout": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "giga_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "giga_host_register": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "giga_host_unregister": (ctypes.c_int, [ctypes.c_void_p]),
    "giga_tsdf_scatter_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "giga_tsdf_scatter": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "giga_train_loss": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                               ctypes.c_void_p]),
    "giga_train_loss_backward": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5),
    "giga_adam_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                      ctypes.c_void_p]),
    "giga_launch_count": (ctypes.c_ulonglong, []),
    "giga_launch_probe": (ctypes.c_int, [ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]),
    "giga_launch_probe_name": (ctypes.c_char_p, []),
    "giga_encoder_last_path": (ctypes.c_int, []),
    "giga_forget_device_state": (None, []),
    "giga_event_create": (ctypes.c_void_p, []),
    "giga_event_destroy": (None, [ctypes.c_void_p]),
    "giga_event_record": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "giga_event_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "giga_encoder_forward_probe": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "giga_decoder_forward_probe": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p]),
    "giga_lattice_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "giga_decoder_forward_lattice": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p]),
    "giga_grasp_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "giga_grasp_select": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)


class GraspParams(ctypes.Structure):
    """struct GigaGraspParams of include/giga_hip.h."""
    _fields_ = [("gaussian_sigma", ctypes.c_double), ("min_width", ctypes.c_float), ("max_width", ctypes.c_float),
                ("out_th", ctypes.c_float), ("low_th", ctypes.c_float), ("threshold", ctypes.c_float),
                ("lim_x", ctypes.c_int), ("lim_y", ctypes.c_int), ("lim_z", ctypes.c_int),
                ("max_filter_size", ctypes.c_int), ("force_detection", ctypes.c_int)]


class GigaHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GigaHipError(
                f"{LIB_PATH} not found: build it with `python -m giga_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.giga_abi_version() != 3:
            raise GigaHipError("libgiga_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise GigaHipError(f"{what} failed: {lib().giga_strerror(code).decode()} ({code})")


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device).  Callers launch inside
    `torch.cuda.device(t.device)` so that the library's kernels, attribute calls and this stream all refer to the device
    the tensors live on, whatever torch's current device is."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def device_of(*tensors):
    """The one HIP device all given tensors live on (GigaHipError for CPU tensors or mixed devices)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            require_device(t)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GigaHipError(f"tensors live on different devices ({dev} and {t.device})")
    return dev


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GigaHipError(
                "giga_amd runs only on a HIP device (tensor is on %s); there is no CPU fallback. "
                "Use the reference PyTorch implementation for CPU execution." % t.device)


def pack_map(head_present):
    """int32 CPU tensor, one entry per 4-byte blob word (see giga_pack_map in include/giga_hip.h)."""
    L = lib()
    n = L.giga_packed_bytes() // 4
    m = torch.empty(n, dtype=torch.int32)
    check(L.giga_pack_map(head_present, ptr(m), n), "giga_pack_map")
    return m


def pack_bwd_map(head_present):
    L = lib()
    n = L.giga_bwd_packed_bytes() // 4
    m = torch.empty(n, dtype=torch.int32)
    check(L.giga_pack_bwd_map(head_present, ptr(m), n), "giga_pack_bwd_map")
    return m


def pack_bwd_weights(flat_params_cpu, head_present):
    L = lib()
    flat = flat_params_cpu.detach().to(dtype=torch.float32, device="cpu").contiguous()
    blob = torch.empty(L.giga_bwd_packed_bytes(), dtype=torch.uint8)
    check(L.giga_pack_bwd_weights(ptr(flat), flat.numel(), head_present, ptr(blob), blob.numel()),
          "giga_pack_bwd_weights")
    return blob


def pack_weights(flat_params_cpu, head_present):
    """flat fp32 CPU tensor (reference state-dict order) -> uint8 CPU tensor (fragment blob)."""
    L = lib()
    flat = flat_params_cpu.detach().to(dtype=torch.float32, device="cpu").contiguous()
    n = L.giga_param_count(head_present)
    if flat.numel() != n:
        raise GigaHipError(f"expected {n} parameters for head set {head_present}, got {flat.numel()}")
    blob = torch.empty(L.giga_packed_bytes(), dtype=torch.uint8)
    check(L.giga_pack_weights(ptr(flat), flat.numel(), head_present, ptr(blob), blob.numel()),
          "giga_pack_weights")
    check(L.giga_packed_check(ptr(blob), blob.numel(), 0), "giga_packed_check")      # (the stamp every uploaded blob must carry)
    return blob
