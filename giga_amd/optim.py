"""Optimiser for the flattened parameters of a giga_amd network.

The reference trains with `torch.optim.Adam(net.parameters(), lr=2e-4)` (scripts/train_giga.py:49) and that keeps working
unchanged.  After `net.flatten_parameters()` all 581 863 weights are ONE tensor, and torch's fused multi-tensor Adam then runs
on nine workgroups (one per 65 536 elements): 98 us of a 1.6-ms training step.  `FlatAdam` is the same update
(torch/optim/adam.py, no amsgrad) as one HIP launch over the whole chip (`giga_adam_step`, ~5 us):

    opt = giga_amd.optim.FlatAdam(net.flatten_parameters(), lr=2e-4)

State-dict layout (`step`, `exp_avg`, `exp_avg_sq` per parameter) is torch.optim.Adam's, so a checkpointed optimiser state moves
between the two.  Device tensors only; no CPU fallback."""
import torch

from . import _capi


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _capi.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _capi.require_device(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FlatAdam updates contiguous fp32 tensors (net.flatten_parameters())")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.data_ptr() % 16:                        # (a gradient view at an odd offset of a larger bucket)
                    g = g.clone()
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)       # (a CPU scalar tensor, as torch's Adam keeps it)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                with torch.cuda.device(p.device):
                    _capi.check(L.giga_adam_step(_capi.ptr(p), _capi.ptr(g), _capi.ptr(st["exp_avg"]), _capi.ptr(st["exp_avg_sq"]),
                                                 p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                 float(group["weight_decay"]), int(st["step"]), _capi.stream_ptr(p.device)),
                                "giga_adam_step")
                # the update went through a raw pointer: tell autograd's version counter, so that caches and stale-graph
                # checks keyed on (storage, version) see it (giga_amd.training, _PackedWeights)
                bump = getattr(torch.autograd.graph, "increment_version", None)
                if bump is not None:
                    bump(p)
        return loss
