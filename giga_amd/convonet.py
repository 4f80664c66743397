"""MI355X-native counterparts of the reference's `vgn.ConvONets` modules on the GIGA hot path.

Same class names, constructor kwargs, `forward` signatures, tensor layouts and state-dict keys as
the reference (paths relative to /root/reference/src/vgn):
  * `LocalVoxelEncoder`               ConvONets/encoder/voxels.py:10-121
  * `UNet`, `DownConv`, `UpConv`      ConvONets/encoder/unet.py:48-239       (parameter containers)
  * `ResnetBlockFC`                   ConvONets/layers.py:6-47               (parameter container)
  * `LocalDecoder`                    ConvONets/conv_onet/models/decoder.py:61-206
  * `ConvolutionalOccupancyNetwork`   ConvONets/conv_onet/models/__init__.py:15-164
  * `ConvolutionalOccupancyNetworkGeometry`                       ...__init__.py:166-226
  * `get_model`                       ConvONets/conv_onet/config.py:15-91

The torch.nn layers below only OWN parameters (so `load_state_dict` of a reference checkpoint works
unchanged and initialisation matches the reference); all arithmetic runs in the hand-written HIP
kernels of libgiga_hip.so through `giga_amd._capi`.  Inputs must live on a HIP device: there is no
CPU path (use the reference for that).  `ConvolutionalOccupancyNetwork.forward` is differentiable
w.r.t. the parameters (fp32 HIP backward, giga_amd/training.py); the piecewise entry points
(`encode_inputs`, `decode`, `LocalDecoder.forward`, ...) are inference-only and raise under autograd.
"""
import os

import torch
import torch.nn as nn
from torch import distributions as dist

from . import _capi

RES = 40
C_DIM = 32
PLANES = ("xz", "xy", "yz")
HEAD_NAMES = ("decoder_qual", "decoder_rot", "decoder_width", "decoder_tsdf")
_DEFAULT_PRECISION = os.environ.get("GIGA_PRECISION", "fp32")


# ------------------------------------------------------------------------------------------------
# parameter containers (names == reference state-dict keys)
# ------------------------------------------------------------------------------------------------
class ResnetBlockFC(nn.Module):
    """layers.py:6-47 (size_in == size_out == size_h: no shortcut).  fc_1.weight zero-init (:37)."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        if not (size_in == size_out == size_h):
            raise NotImplementedError("GIGA uses square ResnetBlockFC only")
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.shortcut = None
        nn.init.zeros_(self.fc_1.weight)


class DownConv(nn.Module):
    """unet.py:48-72."""

    def __init__(self, in_channels, out_channels, pooling=True):
        super().__init__()
        self.in_channels, self.out_channels, self.pooling = in_channels, out_channels, pooling
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)


class UpConv(nn.Module):
    """unet.py:75-114 (merge_mode='concat', up_mode='transpose')."""

    def __init__(self, in_channels, out_channels, merge_mode="concat", up_mode="transpose"):
        super().__init__()
        if merge_mode != "concat" or up_mode != "transpose":
            raise NotImplementedError("GIGA uses concat/transpose UpConv only")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.upconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        self.conv1 = nn.Conv2d(2 * out_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)


class UNet(nn.Module):
    """unet.py:117-239.  Only the GIGA configuration is supported by the kernels:
    UNet(32, in_channels=32, depth=3, start_filts=32, merge_mode='concat')."""

    def __init__(self, num_classes, in_channels=3, depth=5, start_filts=64, up_mode="transpose",
                 merge_mode="concat", **kwargs):
        super().__init__()
        if (num_classes, in_channels, depth, start_filts, up_mode, merge_mode) != \
                (32, 32, 3, 32, "transpose", "concat"):
            raise NotImplementedError("libgiga_hip is specialised for GIGA's U-Net "
                                      "(32->32, depth 3, start_filts 32, transpose/concat)")
        self.num_classes, self.in_channels, self.start_filts, self.depth = \
            num_classes, in_channels, start_filts, depth
        downs, ups = [], []
        outs = in_channels
        for i in range(depth):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            downs.append(DownConv(ins, outs, pooling=i < depth - 1))
        for i in range(depth - 1):
            ins = outs
            outs = ins // 2
            ups.append(UpConv(ins, outs, up_mode=up_mode, merge_mode=merge_mode))
        self.down_convs = nn.ModuleList(downs)
        self.up_convs = nn.ModuleList(ups)
        self.conv_final = nn.Conv2d(outs, num_classes, 1)
        for m in self.modules():                      # unet.py:213-222
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)


class _ScratchCache:
    """Scratch device buffers keyed by (shape key, device, STREAM): a handful of entries in LRU order, so alternating batch
    sizes (a planner at B = 1 next to an evaluation at B = 32) do not reallocate every call, and two streams never share
    a workspace (the library is re-entrant per (stream, workspace): INTEGRATION.md section 4)."""

    def __init__(self, entries=6):
        from collections import OrderedDict
        self._d, self._n = OrderedDict(), entries

    def get(self, key, device, nbytes_fn):
        k = key + (str(device), torch.cuda.current_stream(device).cuda_stream)
        t = self._d.get(k)
        if t is None:
            t = self._d[k] = torch.empty(max(int(nbytes_fn()), 16), dtype=torch.uint8, device=device)
            while len(self._d) > self._n:
                self._d.popitem(last=False)
        else:
            self._d.move_to_end(k)
        return t

    def snapshot(self):
        """The cached buffers (a hipGraph that baked their addresses in keeps them alive with this list)."""
        return list(self._d.values())


class PlaneDict(dict):
    """The reference's {'xz','xy','yz': (B,32,40,40)} dict, plus the NHWC image the HIP decoder
    reads (`nhwc`, `precision`).  Behaves as a plain dict for any reference-style consumer."""
    nhwc = None
    precision = None


# ------------------------------------------------------------------------------------------------
# packed-weight cache
# ------------------------------------------------------------------------------------------------
def _flat_params(named_tensors, device):
    return torch.cat([t.detach().reshape(-1).to(torch.float32) for t in named_tensors]).to(device)


class _PackedWeights:
    """Caches the device blob; repacks when any parameter tensor changes (storage pointer or autograd version counter).
    CAVEAT: an in-place update through `.data` (`p.data.copy_()`, `p.data.mul_()`: EMA, weight clipping, old-style
    loaders) bumps neither -- call `module.invalidate_packed()` after such an update (optimizers, `load_state_dict`,
    `.to()`, `train()`/`eval()` are covered)."""

    def __init__(self):
        self.key = None
        self.blob = None

    def invalidate(self):
        self.key = None
        self.blob = None

    def get(self, params, head_present, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:    # "cuda" and "cuda:0" must share one cache entry
            device = torch.device("cuda", torch.cuda.current_device())
        key = (head_present, str(device)) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self.key:
            flat = _flat_params(params, "cpu")
            self.blob = _capi.pack_weights(flat, head_present).to(device)
            self.key = key
        return self.blob


def _check_no_grad(module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            "the piecewise entry points (encode_inputs / decode / decode_occ / LocalDecoder / LocalVoxelEncoder) are "
            "inference-only: wrap the call in torch.no_grad().  Training goes through the model's own forward, "
            "net(inputs, p, p_tsdf=...), which is differentiable w.r.t. the parameters (DESIGN.md 3b).")


def _head_param_list(dec):
    out = []
    for i in range(5):
        out += [dec.fc_c[i].weight, dec.fc_c[i].bias]
    out += [dec.fc_p.weight, dec.fc_p.bias]
    for i in range(5):
        out += [dec.blocks[i].fc_0.weight, dec.blocks[i].fc_0.bias,
                dec.blocks[i].fc_1.weight, dec.blocks[i].fc_1.bias]
    out += [dec.fc_out.weight, dec.fc_out.bias]
    return out


def _encoder_param_list(enc):
    out = [enc.conv_in.weight, enc.conv_in.bias]
    u = enc.unet
    for d in u.down_convs:
        out += [d.conv1.weight, d.conv1.bias, d.conv2.weight, d.conv2.bias]
    for m in u.up_convs:
        out += [m.upconv.weight, m.upconv.bias, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias]
    out += [u.conv_final.weight, u.conv_final.bias]
    return out


_ENCODER_NUMEL = 476800


# ------------------------------------------------------------------------------------------------
# encoder
# ------------------------------------------------------------------------------------------------
class LocalVoxelEncoder(nn.Module):
    """voxels.py:10-121.  forward(x:(B,40,40,40)) -> {'xz','xy','yz': (B,32,40,40)} (PlaneDict)."""

    def __init__(self, dim=3, c_dim=128, unet=False, unet_kwargs=None, unet3d=False, unet3d_kwargs=None,
                 plane_resolution=512, grid_resolution=None, plane_type="xz", kernel_size=3, padding=0.1):
        super().__init__()
        if not unet or unet3d or c_dim != C_DIM or plane_resolution != RES or kernel_size != 3 \
                or list(plane_type) != list(PLANES) or padding != 0:
            raise NotImplementedError(
                "libgiga_hip is specialised for GIGA's encoder: c_dim=32, unet=True, "
                "plane_resolution=40, plane_type=['xz','xy','yz'], kernel_size=3, padding=0")
        self.conv_in = nn.Conv3d(1, c_dim, kernel_size, padding=1)
        self.unet = UNet(c_dim, in_channels=c_dim, **unet_kwargs)
        self.unet3d = None
        self.c_dim, self.reso_plane, self.reso_grid = c_dim, plane_resolution, grid_resolution
        self.plane_type, self.padding = plane_type, padding
        self.precision = _DEFAULT_PRECISION
        self._packed = _PackedWeights()
        self._ws = _ScratchCache()

    # -- HIP path ---------------------------------------------------------------------------------
    def _blob(self, device, blob=None):
        if blob is not None:
            return blob
        params = _encoder_param_list(self)
        return self._packed.get(params, 0, device)     # encoder-only blob (no heads)

    def encode_nhwc(self, x, blob=None, want_nchw=False, precision=None, probe=None, fold_final=False):
        """Run the HIP encoder.  Returns (nhwc planes [3,B,40,40,32], nchw [3,B,32,40,40] or None).
        probe = (stage, ev_start, ev_stop) brackets one kernel launch with HIP events (bench.py).
        fold_final: stop before conv_final (unet.py:238); the planes are then only valid for `decode_heads(...,
        folded=True)`, whose head images carry that 1x1 convolution inside fc_c (GIGA_FOLD_FINAL, include/giga_hip.h)."""
        _capi.require_device(x)
        prec = _capi.ENCODER_PRECISION[_capi.PRECISION[precision or self.precision]]
        if x.dim() != 4 or tuple(x.shape[1:]) != (RES, RES, RES):
            raise ValueError(f"expected a (B,{RES},{RES},{RES}) TSDF batch, got {tuple(x.shape)}")
        x = x.contiguous().float()
        B = x.shape[0]
        L = _capi.lib()
        blob = self._blob(x.device, blob)
        nhwc = torch.empty((3, B, RES, RES, C_DIM), device=x.device, dtype=_capi.PLANE_DTYPE[prec])
        nchw = torch.empty((3, B, C_DIM, RES, RES), device=x.device, dtype=torch.float32) if want_nchw else None
        ws = self._ws.get((B, prec), x.device, lambda: L.giga_encoder_workspace_bytes(B, prec))
        stage, ev0, ev1 = probe if probe is not None else (-1, None, None)
        if fold_final and want_nchw:
            raise ValueError("fold_final planes are an internal representation; the reference layout needs the final planes")
        with torch.cuda.device(x.device):      # launch on the tensors' device and ITS current stream, whatever torch's current device is
            _capi.check(L.giga_encoder_forward_probe(_capi.ptr(x), _capi.ptr(blob), _capi.ptr(nhwc), _capi.ptr(nchw),
                                                     B, prec | (_capi.FOLD_FINAL if fold_final else 0) |
                                                     {False: 0, True: _capi.PERSIST_UNET, "layers": _capi.LAYERWISE_UNET}[getattr(self, "persistent_unet", False)] |
                                                     ({"conv32": _capi.CONV32_UNET, "conv16": _capi.CONV16_UNET}.get(getattr(self, "unet_kernel", "auto"), 0) if prec in (1, 2) else 0) |
                                                     (_capi.DIRECT_CONV if prec == 0 and getattr(self, "unet_kernel", "auto") == "direct" else 0),
                                                     _capi.ptr(ws), ws.numel(), _capi.stream_ptr(x.device),
                                                     stage, ev0, ev1),
                        "giga_encoder_forward")
        return nhwc, nchw

    def forward(self, x, _blob=None):
        _check_no_grad(self)
        nhwc, nchw = self.encode_nhwc(x, blob=_blob, want_nchw=True)
        fea = PlaneDict((k, nchw[i]) for i, k in enumerate(PLANES))
        fea.nhwc, fea.precision = nhwc, self.precision
        return fea


# ------------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------------
def _planes_to_nhwc(c_plane, precision):
    """PlaneDict fast path, else repack the reference-layout tensors on the device."""
    if isinstance(c_plane, PlaneDict) and c_plane.nhwc is not None and \
            c_plane.nhwc.dtype == _capi.PLANE_DTYPE[_capi.PRECISION[precision]]:
        return c_plane.nhwc
    if list(c_plane.keys()) != list(PLANES):
        raise NotImplementedError("GIGA decoders sample the three planes ['xz','xy','yz']")
    xs = [c_plane[k].contiguous().float() for k in PLANES]
    _capi.require_device(*xs)
    B = xs[0].shape[0]
    for t in xs:
        if tuple(t.shape) != (B, C_DIM, RES, RES):
            raise ValueError(f"expected (B,{C_DIM},{RES},{RES}) planes, got {tuple(t.shape)}")
    prec = _capi.DECODER_PRECISION[_capi.PRECISION[precision]]
    nhwc = torch.empty((3, B, RES, RES, C_DIM), device=xs[0].device, dtype=_capi.PLANE_DTYPE[prec])
    with torch.cuda.device(_capi.device_of(*xs)):
        _capi.check(_capi.lib().giga_planes_pack(_capi.ptr(xs[0]), _capi.ptr(xs[1]), _capi.ptr(xs[2]),
                                                 _capi.ptr(nhwc), B, prec, _capi.stream_ptr(xs[0].device)),
                    "giga_planes_pack")
    return nhwc


# ---- the fixed inference lattice (detection_implicit.py:28-31) -------------------------------------
# A query tensor of shape (1, R^3, 3) that is meshgrid(lin, lin, lin, 'ij') with z fastest is shared by all scenes and
# takes the lattice fast path (giga_decoder_forward_lattice).  `giga_amd.detection.query_lattice()` registers the tensors
# it builds; any other tensor of that shape -- e.g. the reference's own `VGNImplicit.pos`, built by its own constructor and
# handed in through the one-line network switch of INTEGRATION.md -- is recognised FROM ITS DATA, once per tensor object
# and version (one small comparison kernel + a host sync, then cached; negative results are cached too).
_LATTICES = {}          # id(tensor) -> (weakref(tensor), version, (lin (R,) fp32 on the same device, R) or None)
_LATTICE_WS = _ScratchCache()
LATTICE_STATS = {"fast": 0, "generic": 0, "detected": 0}     # decode_heads launches per path (tests, diagnostics)


def _remember_lattice(points, result):
    import weakref
    key = id(points)
    _LATTICES[key] = (weakref.ref(points, lambda _r, k=key: _LATTICES.pop(k, None)), points._version, result)


def register_lattice(points, lin):
    _remember_lattice(points, (lin.to(points.device, torch.float32).contiguous(), int(lin.numel())))
    return points


def _detect_lattice(p):
    """(lin, R) if p (1, R^3, 3) is exactly meshgrid(lin, lin, lin, indexing='ij') flattened with z fastest, else None."""
    n = p.shape[1]
    R = round(n ** (1.0 / 3.0))
    if p.shape[0] != 1 or R < 8 or R > 64 or R * R * R != n or p.dtype != torch.float32 or not p.is_contiguous():
        return None
    g = p[0].view(R, R, R, 3)
    lin = g[:, 0, 0, 0]
    ok = (g[..., 0] == lin[:, None, None]) & (g[..., 1] == lin[None, :, None]) & (g[..., 2] == lin[None, None, :])
    if not bool(ok.all()):                                   # host sync: once per tensor object and version
        return None
    LATTICE_STATS["detected"] += 1
    return lin.clone().contiguous(), R


def _lattice_of(p):
    if not p.is_cuda or p.dim() != 3 or p.shape[0] != 1:
        return None
    ent = _LATTICES.get(id(p))
    if ent is not None and ent[0]() is p and ent[1] == p._version:
        return ent[2]
    res = _detect_lattice(p)
    _remember_lattice(p, res)
    return res


def decode_heads(nhwc, p, blob, head_mask, precision, post, probe=None, folded=False):
    """One fused launch for every head in `head_mask` over p (B,N,3).  Returns dict name->tensor.
    probe = (ev_start, ev_stop) brackets the launch with HIP events (bench.py).
    folded: `nhwc` came from `encode_nhwc(..., fold_final=True)` (planes before conv_final).
    A registered lattice tensor (shape (1, R^3, 3)) is shared by all scenes and takes the lattice path."""
    dev = _capi.device_of(nhwc, p, blob)
    if p.dim() != 3 or p.shape[-1] != 3:
        raise ValueError(f"expected (B,N,3) query points, got {tuple(p.shape)}")
    lat = _lattice_of(p)
    B, N = (nhwc.shape[1] if lat is not None else p.shape[0]), p.shape[1]
    if nhwc.shape[1] != B:
        raise ValueError("batch size of planes and points differ")
    if lat is None:
        p = p.contiguous().float()
    out = {}
    if head_mask & 1:
        out["decoder_qual"] = torch.empty((B, N), device=dev)
    if head_mask & 2:
        out["decoder_rot"] = torch.empty((B, N, 4), device=dev)
    if head_mask & 4:
        out["decoder_width"] = torch.empty((B, N), device=dev)
    if head_mask & 8:
        out["decoder_tsdf"] = torch.empty((B, N), device=dev)
    ev0, ev1 = probe if probe is not None else (None, None)
    LATTICE_STATS["fast" if lat is not None else "generic"] += 1
    if lat is not None:
        lin, R = lat
        prec = _capi.LATTICE_PRECISION[_capi.PRECISION[precision]]
        fold = _capi.FOLD_FINAL if folded else 0
        L = _capi.lib()
        ws = _LATTICE_WS.get((B, R, prec), dev, lambda: L.giga_lattice_workspace_bytes(B, R, prec))
        with torch.cuda.device(dev):
            _capi.check(L.giga_decoder_forward_lattice(
                _capi.ptr(nhwc), _capi.ptr(lin), _capi.ptr(blob), head_mask,
                _capi.ptr(out.get("decoder_qual")), _capi.ptr(out.get("decoder_rot")),
                _capi.ptr(out.get("decoder_width")), _capi.ptr(out.get("decoder_tsdf")),
                B, R, prec | fold, 1 if post else 0, _capi.ptr(ws), ws.numel(), _capi.stream_ptr(dev), ev0, ev1),
                "giga_decoder_forward_lattice")
        return out
    with torch.cuda.device(dev):
        _capi.check(_capi.lib().giga_decoder_forward_probe(
            _capi.ptr(nhwc), _capi.ptr(p), _capi.ptr(blob), head_mask,
            _capi.ptr(out.get("decoder_qual")), _capi.ptr(out.get("decoder_rot")),
            _capi.ptr(out.get("decoder_width")), _capi.ptr(out.get("decoder_tsdf")),
            B, N, _capi.DECODER_PRECISION[_capi.PRECISION[precision]] | (_capi.FOLD_FINAL if folded else 0), 1 if post else 0,
            _capi.stream_ptr(dev), ev0, ev1),
            "giga_decoder_forward")
    return out


class LocalDecoder(nn.Module):
    """decoder.py:61-206.  forward(p:(B,N,3), c_plane:dict) -> (B,N) if out_dim == 1 else (B,N,out_dim)."""

    def __init__(self, dim=3, c_dim=128, hidden_size=256, n_blocks=5, out_dim=1, leaky=False,
                 sample_mode="bilinear", padding=0.1, concat_feat=False, no_xyz=False):
        super().__init__()
        if (dim, c_dim, hidden_size, n_blocks, leaky, sample_mode, padding, concat_feat, no_xyz) != \
                (3, C_DIM, 32, 5, False, "bilinear", 0, True, False) or out_dim not in (1, 4):
            raise NotImplementedError(
                "libgiga_hip is specialised for GIGA's decoders: dim=3, c_dim=32, hidden_size=32, "
                "n_blocks=5, bilinear, padding=0, concat_feat=True, out_dim in {1,4}")
        self.concat_feat, self.c_dim, self.n_blocks = concat_feat, 3 * c_dim, n_blocks
        self.no_xyz, self.hidden_size, self.out_dim = no_xyz, hidden_size, out_dim
        self.fc_c = nn.ModuleList([nn.Linear(self.c_dim, hidden_size) for _ in range(n_blocks)])
        self.fc_p = nn.Linear(dim, hidden_size)
        self.blocks = nn.ModuleList([ResnetBlockFC(hidden_size) for _ in range(n_blocks)])
        self.fc_out = nn.Linear(hidden_size, out_dim)
        self.sample_mode, self.padding = sample_mode, padding
        self.precision = _DEFAULT_PRECISION
        self._packed = _PackedWeights()

    def _standalone(self, device):
        """Blob holding only this head: slot 'rot' for out_dim 4, slot 'tsdf' (raw output) otherwise."""
        slot = 2 if self.out_dim == 4 else 8
        params = _head_param_list(self)
        key_params = params
        pw = self._packed
        key = (slot, str(device)) + tuple((q.data_ptr(), q._version) for q in key_params)
        if key != pw.key:
            flat = torch.cat([_flat_params(params, "cpu"), torch.zeros(_ENCODER_NUMEL)])
            pw.blob = _capi.pack_weights(flat, slot).to(device)
            pw.key = key
        return slot, pw.blob

    def forward(self, p, c_plane, **kwargs):
        _check_no_grad(self)
        _capi.require_device(p)
        nhwc = _planes_to_nhwc(c_plane, self.precision)
        slot, blob = self._standalone(p.device)
        name = "decoder_rot" if slot == 2 else "decoder_tsdf"
        return decode_heads(nhwc, p, blob, slot, self.precision, post=False)[name]


# ------------------------------------------------------------------------------------------------
# full model
# ------------------------------------------------------------------------------------------------
class _ParamListCache:
    """Walking the module tree for the 164 parameters costs ~0.3 ms of Python, as much as the GPU work of one small
    batch.  The tree is static, so the ordered list is built once and dropped whenever torch may have replaced the
    Parameter objects (`_apply`: .to()/.half()/.cuda(); `load_state_dict`, which can assign)."""

    def _ordered_params(self):
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = self.__dict__["_plist"] = self._param_list()
        return pl

    def invalidate_packed(self):
        """Drop every cached derivative of the parameters: the ordered parameter list, the packed weight blobs (module,
        encoder, standalone decoders) and anything keyed on them (the planner's hipGraphs re-capture when the blob
        changes).  Needed only after updates torch cannot see: in-place writes through `.data`, or Parameter objects
        re-assigned on a submodule."""
        self.__dict__.pop("_plist", None)
        for m in self.modules():
            pw = m.__dict__.get("_packed")
            if pw is not None:
                pw.invalidate()
        st = self.__dict__.get("_train_state")
        if st is not None:
            st._wkey = None
        return self

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        if self.__dict__.pop("_flat_param", None) is not None:     # torch is about to replace the storages: un-flatten
            for q in self.parameters():
                q.requires_grad_(True)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if mode != self.training:                            # (training loops call net.train() every iteration: keep that free)
            self.invalidate_packed()
        return super().train(mode)


class ConvolutionalOccupancyNetwork(_ParamListCache, nn.Module):
    """models/__init__.py:15-164: encoder + decoder_qual/rot/width (+ decoder_tsdf)."""

    def __init__(self, decoders, encoder=None, device=None, detach_tsdf=False):
        super().__init__()
        self.decoder_qual = decoders[0].to(device)
        self.decoder_rot = decoders[1].to(device)
        self.decoder_width = decoders[2].to(device)
        if len(decoders) == 4:
            self.decoder_tsdf = decoders[3].to(device)
        self.encoder = encoder.to(device) if encoder is not None else None
        self._device = device
        self.detach_tsdf = detach_tsdf
        self.precision = _DEFAULT_PRECISION
        self._packed = _PackedWeights()

    # -- weights ----------------------------------------------------------------------------------
    def set_precision(self, precision):
        """'fp32' (exact fp32 MFMA, default), 'fp16' (f16 operands, fp32 accumulate: 2-5e-3 on raw logits), 'fp16x3'
        (f16 MFMA on split hi/lo operands in encoder and decoders: fp32-grade results, <= 1e-5, at ~5x the fp32-MFMA rate) or
        'fp16x3+fp16' (the f16x3 encoder under the plain-f16 LATTICE decoder: the throughput decoder without plain f16's encoder error;
        other query sets take the f16x3 decoder);
        'bf16' = the training step's forward arithmetic (bf16 U-Net convolutions, everything else fp32), ~1e-2."""
        if precision not in _capi.PRECISION:
            raise ValueError(precision)
        self.precision = precision
        for m in self.modules():
            if isinstance(m, (LocalDecoder, LocalVoxelEncoder)):
                m.precision = precision
        return self

    def set_persistent_unet(self, enabled=True):
        """How the U-Net layers are launched (include/giga_hip.h, GIGA_PERSIST_UNET / GIGA_LAYERWISE_UNET): False = the default
        (one persistent launch for the whole U-Net -- every batch size in the f16-class modes, from 8 scenes up in fp32 -- with
        the same results as per-layer launches: bit for bit in the f16-class modes, to fp32 rounding in fp32); True = the
        persistent launch also for small fp32 batches; "layers" = one launch per layer.  hipGraph capture is fine in every form.
        At most four persistent launches are in flight per device: the library gives a fifth concurrent call (other streams) one
        launch per layer on its own; other processes sharing the device and concurrent replays of captured graphs are not seen."""
        self.encoder.persistent_unet = "layers" if enabled == "layers" else bool(enabled)
        return self

    def set_unet_kernel(self, kernel="auto"):
        """Which convolution kernels run the f16-class U-Net ('fp16', 'fp16x3'; include/giga_hip.h, GIGA_CONV32_UNET / GIGA_CONV16_UNET):
        "auto" (the library's default: conv32 -- 32x32x16 MFMA register tiles over LDS-resident row bands, same-resolution layer pairs
        fused -- up to 16 scenes, conv16 beyond; the environment variable GIGA_CONV32=0 / 1 overrides), "conv32" or "conv16" (16x16x32,
        wave-private patches; the only kernels of 'fp32' / 'bf16').  A forced kernel makes a scene's result independent of the batch
        size it runs in (bit for bit); "auto" does not across the 16-scene threshold.
        'fp32' only: "direct" keeps the direct 3x3 convolutions (GIGA_DIRECT_CONV) where "auto" runs them as Winograd F(2x2, 3x3) on the
        fp32 MFMA (csrc/giga_wino.h; planes within a few 1e-6 relative of the direct form)."""
        if kernel not in ("auto", "conv16", "conv32", "direct"):
            raise ValueError(kernel)
        self.encoder.unet_kernel = kernel
        return self

    def _head_present(self):
        return 7 | (8 if hasattr(self, "decoder_tsdf") else 0)

    def packed_blob(self, device):
        if self.__dict__.pop("_stale_after_training", False):
            # a training forward ran since the blob was packed.  torch's fused optimizers (Adam(fused=True)) update the
            # parameters WITHOUT bumping their version counters (measured: 0 bumps per step), so the cache key cannot see
            # those steps; the training path therefore marks the inference caches stale itself
            self.invalidate_packed()
        return self._packed.get(self._ordered_params(), self._head_present(), device)

    # -- reference API ------------------------------------------------------------------------------
    def forward(self, inputs, p, p_tsdf=None, sample=True, _probe=None, **kwargs):
        """models/__init__.py:42-67.  inputs (B,40,40,40); p (B,N,3); p_tsdf (B,M,3) ->
        qual (B,N) [sigmoid], rot (B,N,4) [unit], width (B,N) [, tsdf (B,M) raw logits].
        (`_probe`: bench.py's HIP-event bracket around one encoder kernel; not part of the API.)"""
        _capi.require_device(inputs, p, p_tsdf)
        fp = self.__dict__.get("_flat_param")
        if torch.is_grad_enabled() and (fp.requires_grad if fp is not None else any(q.requires_grad for q in self._ordered_params())):
            return self._forward_train(inputs, p, p_tsdf)
        blob = self.packed_blob(inputs.device)
        # encoder and decoders are called back to back here, so conv_final is folded into the heads' fc_c weights and
        # never launched (GIGA_FOLD_FINAL); encode_inputs / decode keep exchanging the final planes, as the reference does
        nhwc, _ = self.encoder.encode_nhwc(inputs, blob=blob, precision=self.precision, probe=_probe, fold_final=True)
        g = decode_heads(nhwc, p, blob, 7, self.precision, post=True, folded=True)
        out = (g["decoder_qual"], g["decoder_rot"], g["decoder_width"])
        if p_tsdf is not None:
            t = decode_heads(nhwc, p_tsdf, blob, 8, self.precision, post=False, folded=True)
            out = out + (t["decoder_tsdf"],)
        return out

    def _param_list(self):
        params = []
        for h in HEAD_NAMES:
            if hasattr(self, h):
                params += _head_param_list(getattr(self, h))
        return params + _encoder_param_list(self.encoder)

    def flatten_parameters(self):
        """Optional fast path for training loops: every parameter becomes a VIEW of one flat fp32 tensor in state-dict order,
        and that tensor is the single trainable leaf:

            opt = torch.optim.Adam(net.flatten_parameters(), lr=2e-4, fused=True)

        The optimizer then updates all 581 863 weights in one launch (the 164-tensor fused Adam takes five launches of 31 us),
        the per-step flattening copy disappears and autograd handles one gradient instead of 164.  `state_dict()`,
        `load_state_dict()` and `named_parameters()` keep their reference keys and shapes (they are the views); the individual
        Parameters stop requiring grad, so build the optimizer from the returned list, not from `net.parameters()`.
        `.to()` / `.float()` undo the flattening (torch replaces the storages).  Returns [flat_parameter]."""
        fp = self.__dict__.get("_flat_param")
        if fp is not None:
            return [fp]
        params = self._ordered_params()
        flat = torch.cat([q.detach().reshape(-1).float() for q in params])
        at = 0
        for q in params:
            n = q.numel()
            q.data = flat[at:at + n].view(q.shape)
            q.requires_grad_(False)
            at += n
        fp = self.__dict__["_flat_param"] = torch.nn.Parameter(flat)
        self.invalidate_packed()
        self.__dict__["_flat_param"] = fp                    # (invalidate_packed keeps it; _apply drops it)
        return [fp]

    def set_train_precision(self, precision):
        """Arithmetic of the differentiable forward/backward (giga_amd/training.py): "fp32" (default), "bf16" (BASELINE config
        c5: bf16 MFMA operands / fp32 accumulation in the U-Net's forward, data-gradient and 3x3 weight-gradient convolutions AND in
        the decoder heads -- forward, gradient chain and weight gradients in one fused kernel per call; conv_in's forward on the f16
        MFMA; fp32 activations in memory, master weights and optimizer) or "bf16_convs" (the convolutions only; fp32 decoders)."""
        if precision not in ("fp32", "bf16", "bf16_convs"):
            raise ValueError(precision)
        self._train_bf16 = precision != "fp32"
        self._train_bf16_dec = precision == "bf16"
        st = getattr(self, "_train_state", None)
        if st is not None:
            st.bf16 = self._train_bf16
            st.bf16_dec = self._train_bf16_dec
            st._wkey = None
        return self

    def enable_data_parallel(self, group=None, enabled=True):
        """Scene-sharded data-parallel training: every rank runs the same step on its own scenes and the
        backward all-reduces (means) the flat gradient bucket once (giga_amd.training.allreduce_mean_)."""
        self._dp = (bool(enabled), group)
        st = getattr(self, "_train_state", None)
        if st is not None:
            st.data_parallel, st.group = self._dp
        return self

    def _forward_train(self, inputs, p, p_tsdf):
        """Differentiable fp32 path (scripts/train_giga.py:204): HIP forward + HIP backward through
        giga_amd.training.GigaFunction.  Gradients flow to the parameters only."""
        from .training import GigaFunction, _TrainState
        st = getattr(self, "_train_state", None)
        if st is None or st.blob.device != inputs.device:
            st = self._train_state = _TrainState(self._head_present(), inputs.device, detach_occ=self.detach_tsdf)
            st.data_parallel, st.group = getattr(self, "_dp", (False, None))
            st.bf16 = getattr(self, "_train_bf16", False)
            st.bf16_dec = getattr(self, "_train_bf16_dec", False)
        self.__dict__["_stale_after_training"] = True
        fp = self.__dict__.get("_flat_param")
        if fp is not None:
            return GigaFunction.apply(st, inputs, p, p_tsdf, fp)
        return GigaFunction.apply(st, inputs, p, p_tsdf, *self._ordered_params())

    def infer_geo(self, inputs, p_tsdf, **kwargs):
        """models/__init__.py:69-72."""
        c = self.encode_inputs(inputs)
        return self._decode_tsdf(p_tsdf, c)

    def encode_inputs(self, inputs):
        """models/__init__.py:74-87."""
        if self.encoder is None:
            return torch.empty(inputs.size(0), 0)
        _check_no_grad(self)
        nhwc, nchw = self.encoder.encode_nhwc(inputs, blob=self.packed_blob(inputs.device), want_nchw=True,
                                              precision=self.precision)
        fea = PlaneDict((k, nchw[i]) for i, k in enumerate(PLANES))
        fea.nhwc, fea.precision = nhwc, self.precision
        return fea

    def _decode_tsdf(self, p, c):
        blob = self.packed_blob(p.device)
        nhwc = _planes_to_nhwc(c, self.precision)
        return decode_heads(nhwc, p, blob, 8, self.precision, post=False)["decoder_tsdf"]

    def decode_occ(self, p, c, **kwargs):
        """models/__init__.py:100-109."""
        _check_no_grad(self)
        return dist.Bernoulli(logits=self._decode_tsdf(p, c))

    def decode(self, p, c, **kwargs):
        """models/__init__.py:111-124."""
        _check_no_grad(self)
        _capi.require_device(p)
        blob = self.packed_blob(p.device)
        nhwc = _planes_to_nhwc(c, self.precision)
        g = decode_heads(nhwc, p, blob, 7, self.precision, post=True)
        return g["decoder_qual"], g["decoder_rot"], g["decoder_width"]

    def to(self, device):
        """models/__init__.py:126-134."""
        model = super().to(device)
        model._device = device
        return model

    def grad_refine(self, x, pos, bound_value=0.0125, lr=1e-6, num_step=1):
        """models/__init__.py:136-164 optimises the query POSITIONS by gradient ascent on the predicted quality; it needs
        d qual / d p, which the HIP decoder does not produce (its backward yields parameter gradients only)."""
        raise NotImplementedError("grad_refine needs gradients with respect to the query points; libgiga_hip's backward "
                                  "computes parameter gradients only (DESIGN.md, out of SURVEY 8's scope)")


class ConvolutionalOccupancyNetworkGeometry(_ParamListCache, nn.Module):
    """models/__init__.py:166-226 (occupancy head only; `giga_geo`)."""

    def __init__(self, decoder, encoder=None, device=None):
        super().__init__()
        self.decoder_tsdf = decoder.to(device)
        self.encoder = encoder.to(device) if encoder is not None else None
        self._device = device
        self.precision = _DEFAULT_PRECISION
        self._packed = _PackedWeights()

    def set_precision(self, precision):
        if precision not in _capi.PRECISION:
            raise ValueError(precision)
        self.precision = precision
        self.decoder_tsdf.precision = precision
        self.encoder.precision = precision
        return self

    def _param_list(self):
        return _head_param_list(self.decoder_tsdf) + _encoder_param_list(self.encoder)

    def packed_blob(self, device):
        return self._packed.get(self._ordered_params(), 8, device)

    def forward(self, inputs, p, p_tsdf, sample=True, **kwargs):
        _check_no_grad(self)
        _capi.require_device(inputs, p_tsdf)
        blob = self.packed_blob(inputs.device)
        nhwc, _ = self.encoder.encode_nhwc(inputs, blob=blob, precision=self.precision)
        return decode_heads(nhwc, p_tsdf, blob, 8, self.precision, post=False)["decoder_tsdf"]

    def infer_geo(self, inputs, p_tsdf, **kwargs):
        return self.forward(inputs, None, p_tsdf)

    def encode_inputs(self, inputs):
        _check_no_grad(self)
        nhwc, nchw = self.encoder.encode_nhwc(inputs, blob=self.packed_blob(inputs.device), want_nchw=True,
                                              precision=self.precision)
        fea = PlaneDict((k, nchw[i]) for i, k in enumerate(PLANES))
        fea.nhwc, fea.precision = nhwc, self.precision
        return fea

    def decode_occ(self, p, c, **kwargs):
        _check_no_grad(self)
        nhwc = _planes_to_nhwc(c, self.precision)
        logits = decode_heads(nhwc, p, self.packed_blob(p.device), 8, self.precision, post=False)["decoder_tsdf"]
        return dist.Bernoulli(logits=logits)

    def to(self, device):
        model = super().to(device)
        model._device = device
        return model


decoder_dict = {"simple_local": LocalDecoder}          # models/__init__.py:7-12 (GIGA entry only)
encoder_dict = {"voxel_simple_local": LocalVoxelEncoder}   # encoder/__init__.py:6-11 (GIGA entry only)


def get_model(cfg, device=None, dataset=None, **kwargs):
    """conv_onet/config.py:15-91 for the dict configs of networks.py:65-169."""
    decoder, encoder = cfg["decoder"], cfg["encoder"]
    c_dim = cfg["c_dim"]
    decoder_kwargs, encoder_kwargs = dict(cfg["decoder_kwargs"]), dict(cfg["encoder_kwargs"])
    padding = cfg["padding"]
    if padding is None:
        padding = 0.1
    tsdf_only = bool(cfg.get("tsdf_only"))
    detach_tsdf = bool(cfg.get("detach_tsdf"))
    decoders = []
    if not tsdf_only:
        for out_dim in (1, 4, 1):
            decoders.append(decoder_dict[decoder](c_dim=c_dim, padding=padding, out_dim=out_dim, **decoder_kwargs))
    if cfg["decoder_tsdf"] or tsdf_only:
        decoder_tsdf = decoder_dict[decoder](c_dim=c_dim, padding=padding, out_dim=1, **decoder_kwargs)
        decoders.append(decoder_tsdf)
    enc = encoder_dict[encoder](c_dim=c_dim, padding=padding, **encoder_kwargs) if encoder is not None else None
    if tsdf_only:
        return ConvolutionalOccupancyNetworkGeometry(decoder_tsdf, enc, device=device)
    return ConvolutionalOccupancyNetwork(decoders, enc, device=device, detach_tsdf=detach_tsdf)
