"""Training path: autograd bridge to the HIP forward/backward kernels (fp32) and the counterparts of the
training caller's helpers (reference scripts/train_giga.py:141-218).

`ConvolutionalOccupancyNetwork.forward` dispatches here when autograd is enabled and parameters require
grad, so the reference loop works unchanged:

    y_pred = select(net(x, pos, p_tsdf=pos_occ)); loss, _ = loss_fn(y_pred, y); loss.backward(); optimizer.step()

Every step the fp32 weight images (forward fragments and the transposed/flipped backward fragments) are
rebuilt ON THE DEVICE from the current parameters with a gather map (giga_repack_device); activations of
the forward stay in the encoder workspace that the backward consumes; gradients come back as one flat
buffer in state-dict order and are handed to autograd as per-parameter views."""
import ctypes

import torch
import torch.nn.functional as F

from . import _capi

RES, C_DIM = 40, 32


class _TrainState:
    """Per-module device state: gather maps and the two weight images."""

    def __init__(self, head_present, device, detach_occ=False):
        self.bwd_flags = _capi.DETACH_OCC if detach_occ else 0     # detach_tsdf (models/__init__.py:61-63)
        L = _capi.lib()
        self.head_present = head_present
        self.map_fwd = _capi.pack_map(head_present).to(device)
        self.map_bwd = _capi.pack_bwd_map(head_present).to(device)
        self.blob = torch.zeros(L.giga_packed_bytes(), dtype=torch.uint8, device=device)
        self.bwd_blob = torch.zeros(L.giga_bwd_packed_bytes(), dtype=torch.uint8, device=device)
        self.n_params = L.giga_param_count(head_present)
        self.data_parallel = False      # set by ConvolutionalOccupancyNetwork.enable_data_parallel()
        self.group = None

    def repack(self, flat):
        L = _capi.lib()
        _capi.check(L.giga_repack_device(_capi.ptr(flat), _capi.ptr(self.map_fwd), _capi.ptr(self.blob),
                                         self.map_fwd.numel(), _capi.stream_ptr()), "giga_repack_device")
        _capi.check(L.giga_repack_device(_capi.ptr(flat), _capi.ptr(self.map_bwd), _capi.ptr(self.bwd_blob),
                                         self.map_bwd.numel(), _capi.stream_ptr()), "giga_repack_device")


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


class GigaFunction(torch.autograd.Function):
    """(x, p, p_tsdf, *params) -> (qual, rot, width[, tsdf]) on the HIP kernels, differentiable w.r.t. params."""

    @staticmethod
    def forward(ctx, state, x, p, p_tsdf, *params):
        from .convonet import decode_heads
        L = _capi.lib()
        dev = x.device
        x = x.contiguous().float()
        p = p.contiguous().float()
        p_tsdf = p_tsdf.contiguous().float() if p_tsdf is not None else None
        B, N = p.shape[0], p.shape[1]
        M = p_tsdf.shape[1] if p_tsdf is not None else 0
        flat = torch.cat([q.detach().reshape(-1).float() for q in params])
        if flat.numel() != state.n_params:
            raise _capi.GigaHipError("parameter list does not match the head set")
        state.repack(flat)
        ws = torch.empty(max(L.giga_encoder_workspace_bytes(B, 0), 16), dtype=torch.uint8, device=dev)
        nhwc = torch.empty((3, B, RES, RES, C_DIM), device=dev, dtype=torch.float32)
        _capi.check(L.giga_encoder_forward(_capi.ptr(x), _capi.ptr(state.blob), _capi.ptr(nhwc), None, B, 0,
                                           _capi.ptr(ws), ws.numel(), _capi.stream_ptr()), "giga_encoder_forward")
        grasp_mask = state.head_present & 7
        outs = [None, None, None, None]
        if grasp_mask:
            g = decode_heads(nhwc, p, state.blob, grasp_mask, "fp32", True)
            outs[0], outs[1], outs[2] = g.get("decoder_qual"), g.get("decoder_rot"), g.get("decoder_width")
        if state.head_present & 8 and p_tsdf is not None:
            outs[3] = decode_heads(nhwc, p_tsdf, state.blob, 8, "fp32", False)["decoder_tsdf"]
        ctx.state, ctx.dims = state, (B, N, M)
        # save_for_backward, not a plain attribute: the head outputs are needed by the backward, and a node that holds
        # its own outputs in a Python attribute is a reference cycle -- the 207 MB activation workspace would then live
        # until the cyclic GC runs (GBs of growth, a hipMalloc every other step and a ~100 ms collection pause)
        ctx.save_for_backward(x, p, p_tsdf, ws, nhwc, *outs)
        ctx.shapes = [tuple(q.shape) for q in params]
        result = tuple(o for o in outs if o is not None)
        ctx.out_slots = [i for i, o in enumerate(outs) if o is not None]
        return result

    @staticmethod
    def backward(ctx, *grad_outs):
        L = _capi.lib()
        state = ctx.state
        B, N, M = ctx.dims
        x, p, p_tsdf, ws, nhwc, *outs = ctx.saved_tensors
        dev = x.device
        douts = [None, None, None, None]
        for slot, gout in zip(ctx.out_slots, grad_outs):
            douts[slot] = (gout if gout is not None else torch.zeros_like(outs[slot])).contiguous().float()
        grads = torch.empty(state.n_params, device=dev, dtype=torch.float32)
        wsb = torch.empty(L.giga_backward_workspace_bytes(B, N, M, state.head_present), dtype=torch.uint8, device=dev)
        _capi.check(L.giga_backward(
            _capi.ptr(x), _capi.ptr(state.blob), _capi.ptr(state.bwd_blob), _capi.ptr(ws), _capi.ptr(nhwc),
            _capi.ptr(p), _capi.ptr(p_tsdf), _ptr_array(outs), _ptr_array(douts), _capi.ptr(grads),
            grads.numel(), state.head_present | state.bwd_flags, B, N, M, _capi.ptr(wsb), wsb.numel(), _capi.stream_ptr()),
            "giga_backward")
        if state.data_parallel:
            allreduce_mean_(grads, state.group)
        views, at = [], 0
        for shp in ctx.shapes:
            n = 1
            for s in shp:
                n *= s
            views.append(grads[at:at + n].view(shp))
            at += n
        return (None, None, None, None) + tuple(views)


# ---------------------------------------------------------------------------------------------------------
# caller-side helpers: scripts/train_giga.py:141-195 (plain torch on the device; they touch only the head
# outputs, i.e. O(B) and O(B*M) elementwise work)
# ---------------------------------------------------------------------------------------------------------
def prepare_batch(batch, device):
    """train_giga.py:141-151."""
    pc, (label, rotations, width), pos, pos_occ, occ_value = batch
    pc = pc.float().to(device)
    label = label.float().to(device)
    rotations = rotations.float().to(device)
    width = width.float().to(device)
    pos = pos.unsqueeze(1).float().to(device)          # B, 1, 3
    pos_occ = pos_occ.float().to(device)
    occ_value = occ_value.float().to(device)
    return pc, (label, rotations, width, occ_value), pos, pos_occ


def select(out):
    """train_giga.py:154-158."""
    qual_out, rot_out, width_out, occ = out
    return qual_out.squeeze(-1), rot_out.squeeze(1), width_out.squeeze(-1), torch.sigmoid(occ)


def _quat_loss_fn(pred, target):
    return 1.0 - torch.abs(torch.sum(pred * target, dim=1))


def loss_fn(y_pred, y):
    """train_giga.py:161-195."""
    label_pred, rotation_pred, width_pred, occ_pred = y_pred
    label, rotations, width, occ = y
    loss_qual = F.binary_cross_entropy(label_pred, label, reduction="none")
    loss_rot = torch.min(_quat_loss_fn(rotation_pred, rotations[:, 0]), _quat_loss_fn(rotation_pred, rotations[:, 1]))
    loss_width = F.mse_loss(40 * width_pred, 40 * width, reduction="none")
    loss_occ = F.binary_cross_entropy(occ_pred, occ, reduction="none").mean(-1)
    loss = loss_qual + label * (loss_rot + 0.01 * loss_width) + loss_occ
    loss_dict = {"loss_qual": loss_qual.mean(), "loss_rot": loss_rot.mean(), "loss_width": loss_width.mean(),
                 "loss_occ": loss_occ.mean(), "loss_all": loss.mean()}
    return loss.mean(), loss_dict
