"""Training path: autograd bridge to the HIP forward/backward kernels and the fused loss of the joint objective
(reference scripts/train_giga.py:154-211).  Arithmetic modes (`net.set_train_precision`): "fp32" (every GEMM on the
fp32-input MFMA; gradients match torch autograd to <= 6e-6 relative), "bf16" (BASELINE config c5: the forward, data-gradient
and weight-gradient GEMMs of the 3x3 U-Net layers AND the decoder heads -- forward, gradient chain and weight gradients in one fused
kernel per call, csrc/giga_decoder_train16.hip -- take bf16 operands with fp32 accumulation on the bf16 MFMA; activations in memory,
conv_in's backward (its forward runs on the f16 MFMA: 11-bit operands), ConvTranspose / 1x1 weight gradients, master weights and the optimizer stay fp32, as under
torch.autocast; gradients then carry bf16 operand rounding) and "bf16_convs" (the convolutions only; fp32 decoders).
The backward runs its weight gradients on a second, library-owned stream beside the data-gradient chain and joins it back into the
caller's stream before it returns (csrc/giga_side.h).

`ConvolutionalOccupancyNetwork.forward` dispatches here when autograd is enabled and parameters require grad, so the
reference loop works unchanged with ITS OWN helpers (`select`, `loss_fn` of train_giga.py stay in the caller):

    y_pred = select(net(x, pos, p_tsdf=pos_occ)); loss, _ = loss_fn(y_pred, y); loss.backward(); optimizer.step()

and `giga_loss(net(x, pos, p_tsdf=pos_occ), y)` is the fused replacement of `loss_fn(select(...), y)`: three HIP launches
instead of ~40 ATen kernels (csrc/giga_loss.hip).

Every step the fp32 weight images (forward fragments and the transposed/flipped backward fragments) are rebuilt ON THE
DEVICE from the current parameters with a gather map (giga_repack_device); the activations of the forward stay in an
encoder workspace that the backward consumes; gradients come back as one flat buffer in state-dict order and are handed
to autograd as per-parameter views.  The large device buffers of a step (activation workspace, planes, backward
workspace) are owned by the module's `_TrainState` and recycled, so a steady-state step only takes the small head outputs
and the 2.3 MB gradient bucket from torch's caching allocator."""
import ctypes

import torch

from . import _capi

RES, C_DIM = 40, 32


class _StepBuffers:
    """The large device buffers of one forward/backward pair for a fixed (B, N, M): activation workspace (207 MB at
    B = 32), planes, backward workspace.  `generation` counts how often the set was handed out, so a backward can tell
    that its activations were recycled by a later forward.  (The head outputs and the flat gradient buffer are NOT
    recycled: callers keep the outputs, and autograd's AccumulateGrad adopts the gradient views as `param.grad`.)"""

    def __init__(self, state, B, N, M, dev):
        L = _capi.lib()
        hp = state.head_present
        self.key = (B, N, M)
        self.busy = False
        self.generation = 0
        self.ws = torch.empty(max(L.giga_encoder_workspace_bytes(B, 0), 16), dtype=torch.uint8, device=dev)
        self.nhwc = torch.empty((3, B, RES, RES, C_DIM), device=dev, dtype=torch.float32)
        self.wsb = torch.empty(max(L.giga_backward_workspace_bytes(B, N, M, hp), 16), dtype=torch.uint8, device=dev)


class _TrainState:
    """Per-module device state: gather maps, the two weight images and the recycled step buffers."""

    def __init__(self, head_present, device, detach_occ=False):
        self.bwd_flags = _capi.DETACH_OCC if detach_occ else 0     # detach_tsdf (models/__init__.py:61-63)
        L = _capi.lib()
        self.head_present = head_present
        self.device = device
        self.map_fwd = _capi.pack_map(head_present).to(device)
        self.map_bwd = _capi.pack_bwd_map(head_present).to(device)
        self.blob = torch.zeros(L.giga_packed_bytes(), dtype=torch.uint8, device=device)
        self.bwd_blob = torch.zeros(L.giga_bwd_packed_bytes(), dtype=torch.uint8, device=device)
        self.n_params = L.giga_param_count(head_present)
        self.flat = torch.empty(self.n_params, device=device, dtype=torch.float32)
        self.repacks = 0                # how often the images were rebuilt: identifies the weights they currently hold
        self._wkey = None               # (storage, version) of every parameter the images were built from
        self.bf16 = False               # set by ConvolutionalOccupancyNetwork.set_train_precision("bf16" / "bf16_convs")
        self.bf16_dec = False           # ... ("bf16"): the decoder heads on the fused bf16 kernels (csrc/giga_decoder_train16.hip)
        self.data_parallel = False      # set by ConvolutionalOccupancyNetwork.enable_data_parallel()
        self.group = None
        self._pool = []

    def repack(self, params):
        """Flatten the parameters (state-dict order) and rebuild both fp32 weight images from them on the device.
        Unconditional (two small gather kernels): correctness of a step never depends on torch's version counters, which
        in-place updates through `.data` do not bump.  `_wkey` (storage + version of every parameter) only serves the
        stale-graph check in GigaFunction.backward."""
        key = tuple((q.data_ptr(), q._version) for q in params)
        L = _capi.lib()
        if len(params) == 1 and params[0].dim() == 1:       # flattened module (flatten_parameters): the parameter IS the buffer
            flat = params[0].detach()
            if flat.numel() != self.n_params or flat.dtype != torch.float32 or not flat.is_contiguous():
                raise _capi.GigaHipError("flat parameter does not match the head set")
        else:
            flat = self.flat
            views, at = [], 0
            for q in params:
                n = q.numel()
                views.append(flat[at:at + n].view(q.shape))
                at += n
            if at != self.n_params:
                raise _capi.GigaHipError("parameter list does not match the head set")
            torch._foreach_copy_(views, [q.detach() for q in params])
        s = _capi.stream_ptr(self.device)
        _capi.check(L.giga_repack_device2(_capi.ptr(flat), _capi.ptr(self.map_fwd), _capi.ptr(self.blob), self.map_fwd.numel(),
                                          _capi.ptr(self.map_bwd), _capi.ptr(self.bwd_blob), self.map_bwd.numel(), s),
                    "giga_repack_device2")
        if self.bf16:                   # bf16 images of the convolution fragments, from the fp32 fragments just rebuilt
            _capi.check(L.giga_derive_bf16_fragments(_capi.ptr(self.blob), _capi.ptr(self.bwd_blob), s),
                        "giga_derive_bf16_fragments")
        fwd32 = (_capi.ENC_BF16 if self.bf16 else 0) == 0          # an fp32 forward / an fp32 data-gradient chain run their 3x3 layers as
        if fwd32 or not self.bf16:                                 # Winograd F(2x2, 3x3): their images, from the fragments just rebuilt
            _capi.check(L.giga_derive_winograd(_capi.ptr(self.blob) if fwd32 else None,
                                               _capi.ptr(self.bwd_blob) if not self.bf16 else None, s), "giga_derive_winograd")
        if key != self._wkey:
            self.repacks += 1
        self._wkey = key

    def acquire(self, B, N, M):
        """A free buffer set for (B, N, M); sets of other shapes are dropped (one shape per training run is the norm)."""
        for sb in self._pool:
            if sb.key == (B, N, M) and not sb.busy:
                break
        else:
            self._pool = [sb for sb in self._pool if sb.key == (B, N, M)]
            sb = _StepBuffers(self, B, N, M, self.device)
            self._pool.append(sb)
        sb.busy = True
        sb.generation += 1
        return sb


def allreduce_mean_(flat, group=None):
    """The single collective of data-parallel training: one all_reduce(sum) over the flat
    581 863-element fp32 gradient bucket (2.3 MB; latency-bound over xGMI), then divide by the world size.
    No-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * 4)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class _Lease:
    """Held by the autograd node only: when the graph dies without a backward (an evaluation under enable_grad, an
    exception), the buffer set returns to the pool instead of staying marked busy."""

    def __init__(self, sb):
        self.sb, self.generation = sb, sb.generation

    def release(self):
        if self.sb is not None and self.sb.generation == self.generation:
            self.sb.busy = False
        self.sb = None

    __del__ = release


class GigaFunction(torch.autograd.Function):
    """(x, p, p_tsdf, *params) -> (qual, rot, width[, tsdf]) on the HIP kernels, differentiable w.r.t. params."""

    @staticmethod
    def forward(ctx, state, x, p, p_tsdf, *params):
        L = _capi.lib()
        dev = x.device
        x = x.contiguous().float()
        p = p.contiguous().float()
        p_tsdf = p_tsdf.contiguous().float() if p_tsdf is not None else None
        B, N = p.shape[0], p.shape[1]
        M = p_tsdf.shape[1] if p_tsdf is not None else 0
        with torch.cuda.device(dev):
            state.repack(params)
            sb = state.acquire(B, N, M)
            s = _capi.stream_ptr(dev)
            _capi.check(L.giga_encoder_forward(_capi.ptr(x), _capi.ptr(state.blob), _capi.ptr(sb.nhwc), None, B,
                                               (_capi.ENC_BF16 if state.bf16 else 0) | _capi.CONVIN_MASK,
                                               _capi.ptr(sb.ws), sb.ws.numel(), s), "giga_encoder_forward")
            hp = state.head_present
            o = [torch.empty((B, N), device=dev) if hp & 1 and N > 0 else None,
                 torch.empty((B, N, 4), device=dev) if hp & 2 and N > 0 else None,
                 torch.empty((B, N), device=dev) if hp & 4 and N > 0 else None,
                 torch.empty((B, M), device=dev) if hp & 8 and M > 0 else None]
            dprec = _capi.DEC_BF16 if state.bf16 and state.bf16_dec else 0
            if state.head_present & 7 and N > 0:
                _capi.check(L.giga_decoder_forward(_capi.ptr(sb.nhwc), _capi.ptr(p), _capi.ptr(state.blob),
                                                   state.head_present & 7, _capi.ptr(o[0]), _capi.ptr(o[1]), _capi.ptr(o[2]),
                                                   None, B, N, dprec, 1, s), "giga_decoder_forward")
            if state.head_present & 8 and M > 0:
                _capi.check(L.giga_decoder_forward(_capi.ptr(sb.nhwc), _capi.ptr(p_tsdf), _capi.ptr(state.blob), 8, None,
                                                   None, None, _capi.ptr(o[3]), B, M, dprec, 0, s), "giga_decoder_forward")
        ctx.state, ctx.dims, ctx.lease, ctx.repacks = state, (B, N, M), _Lease(sb), state.repacks
        ctx.bf16 = state.bf16
        ctx.bf16_dec = state.bf16 and state.bf16_dec
        # save_for_backward, not a plain attribute: a node that holds its own outputs in a Python attribute is a reference
        # cycle (output -> grad_fn -> ctx -> output) that only the cyclic GC would free
        ctx.save_for_backward(x, p, p_tsdf, *o)
        ctx.shapes = [tuple(q.shape) for q in params]
        ctx.out_slots = [i for i, t in enumerate(o) if t is not None]
        return tuple(o[i] for i in ctx.out_slots)

    @staticmethod
    def backward(ctx, *grad_outs):
        L = _capi.lib()
        state = ctx.state
        B, N, M = ctx.dims
        lease = ctx.lease
        sb = lease.sb
        if sb is None or sb.generation != lease.generation:
            raise RuntimeError("giga_amd: the activations of this forward were recycled (its backward already ran, or "
                               "its buffers were released); re-run the forward")
        if state.repacks != ctx.repacks:
            raise RuntimeError("giga_amd: another training forward has re-packed the weight images since this forward ran "
                               "(interleaved forwards with different weights are not supported); backward it first")
        x, p, p_tsdf, *outs = ctx.saved_tensors
        dev = x.device
        douts = [None, None, None, None]
        for slot, gout in zip(ctx.out_slots, grad_outs):
            douts[slot] = (gout if gout is not None else torch.zeros_like(outs[slot])).contiguous().float()
        grads = torch.empty(state.n_params, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _capi.check(L.giga_backward(
                _capi.ptr(x), _capi.ptr(state.blob), _capi.ptr(state.bwd_blob), _capi.ptr(sb.ws), _capi.ptr(sb.nhwc),
                _capi.ptr(p), _capi.ptr(p_tsdf), _ptr_array(outs), _ptr_array(douts), _capi.ptr(grads),
                grads.numel(), state.head_present | state.bwd_flags | (_capi.BF16_CONVS if ctx.bf16 else 0) |
                (_capi.BF16_DECODER if ctx.bf16_dec else 0) | _capi.CONVIN_MASK_BWD, B, N, M,
                _capi.ptr(sb.wsb), sb.wsb.numel(),
                _capi.stream_ptr(dev)), "giga_backward")
        if state.data_parallel:
            allreduce_mean_(grads, state.group)
        lease.release()
        if len(ctx.shapes) == 1 and len(ctx.shapes[0]) == 1:  # flattened module: one gradient for the one flat parameter
            return (None, None, None, None, grads)
        views, at = [], 0
        for shp in ctx.shapes:
            n = 1
            for d in shp:
                n *= d
            views.append(grads[at:at + n].view(shp))
            at += n
        return (None, None, None, None) + tuple(views)


# ---------------------------------------------------------------------------------------------------------
# fused loss: the counterpart of loss_fn(select(out), y) (scripts/train_giga.py:154-195) on the device
# ---------------------------------------------------------------------------------------------------------
LOSS_KEYS = ("loss_qual", "loss_rot", "loss_width", "loss_occ", "loss_all")      # train_giga.py:169-173


class _GigaLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qual, rot, width, occ, label, rot_t, width_t, occ_t):
        L = _capi.lib()
        B, M = occ.shape[0], occ.shape[1]
        dev = occ.device
        args = [t.contiguous().float() for t in (qual, rot, width, occ, label, rot_t, width_t, occ_t)]
        losses = torch.empty(5, device=dev)
        scene = torch.empty((B, 5), device=dev)
        with torch.cuda.device(dev):
            _capi.check(L.giga_train_loss(*[_capi.ptr(t) for t in args], B, M, _capi.ptr(losses), _capi.ptr(scene),
                                          _capi.stream_ptr(dev)), "giga_train_loss")
        ctx.save_for_backward(*args)
        ctx.shapes = (qual.shape, rot.shape, width.shape, occ.shape)
        parts = losses[:4]
        ctx.mark_non_differentiable(parts)
        return losses[4], parts

    @staticmethod
    def backward(ctx, gloss, _gparts):
        L = _capi.lib()
        args = ctx.saved_tensors
        occ = args[3]
        B, M = occ.shape[0], occ.shape[1]
        dev = occ.device
        g = gloss.contiguous().float().reshape(1)
        dq, dr, dw, do = (torch.empty(s, device=dev) for s in ctx.shapes)
        with torch.cuda.device(dev):
            _capi.check(L.giga_train_loss_backward(*[_capi.ptr(t) for t in args], _capi.ptr(g), B, M, _capi.ptr(dq),
                                                   _capi.ptr(dr), _capi.ptr(dw), _capi.ptr(do), _capi.stream_ptr(dev)),
                        "giga_train_loss_backward")
        return dq, dr, dw, do, None, None, None, None


def giga_loss(out, y):
    """Fused `loss_fn(select(out), y)` of scripts/train_giga.py:154-195 for the literal call shape.

    out: the model's 4-tuple for ONE grasp query per scene -- qual (B,1), rot (B,1,4), width (B,1), occupancy logits (B,M);
    y:   (label (B,), rotations (B,2,4), width (B,), occ (B,M)) as `prepare_batch` builds them (train_giga.py:141-151).
    Returns (loss, loss_dict) with the reference's keys; `loss` is differentiable w.r.t. the four head outputs."""
    qual, rot, width, occ = out
    label, rot_t, width_t, occ_t = y
    _capi.require_device(qual, rot, width, occ, label, rot_t, width_t, occ_t)
    if occ.dim() != 2:
        raise ValueError(f"expected occupancy logits of shape (B, M), got {tuple(occ.shape)}")
    B, M = occ.shape
    if qual.numel() != B or rot.numel() != 4 * B or width.numel() != B:
        raise ValueError("giga_loss is the fused form of the literal train_giga call: one grasp query per scene")
    for name, t, shape in (("label", label, (B,)), ("rotations", rot_t, (B, 2, 4)), ("width", width_t, (B,)), ("occ", occ_t, (B, M))):
        if tuple(t.shape) != shape:                          # (the kernels index the targets by these shapes)
            raise ValueError(f"giga_loss: target `{name}` has shape {tuple(t.shape)}, expected {shape}")
    if B == 0:                                               # F.*_loss(...).mean() of an empty batch
        nan = occ.new_full((), float("nan")) + 0.0 * occ.sum()
        return nan, {k: nan.detach() for k in LOSS_KEYS}
    loss, parts = _GigaLoss.apply(qual, rot, width, occ, label, rot_t, width_t, occ_t)
    d = {k: parts[i] for i, k in enumerate(LOSS_KEYS[:4])}
    d["loss_all"] = loss.detach()
    return loss, d
