"""Host side of giga_amd.optim.FlatAdam (no GPU): argument checks, and that there is no CPU path behind it."""
import pytest
import torch

from giga_amd import _capi
from giga_amd.optim import FlatAdam


def test_flat_adam_rejects_bad_hyper_parameters():
    p = torch.nn.Parameter(torch.zeros(8))
    for kw in (dict(lr=-1.0), dict(eps=-1e-8), dict(weight_decay=-0.1), dict(betas=(1.0, 0.999)), dict(betas=(0.9, -0.1))):
        with pytest.raises(ValueError):
            FlatAdam([p], **kw)
    opt = FlatAdam([p], lr=2e-4)
    assert opt.defaults == dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)


def test_flat_adam_has_no_cpu_fallback():
    p = torch.nn.Parameter(torch.zeros(8))
    opt = FlatAdam([p], lr=2e-4)
    opt.step()                                   # no gradient yet: nothing to do, no state
    assert not opt.state
    p.grad = torch.ones(8)
    with pytest.raises(_capi.GigaHipError):      # host tensors are refused, the parameter is untouched
        opt.step()
    assert torch.equal(p.detach(), torch.zeros(8))


def test_adam_step_abi_validates_before_it_launches():
    lib = _capi.lib()
    fake = torch.zeros(4)
    ptr = _capi.ptr(fake)
    assert lib.giga_adam_step(None, ptr, ptr, ptr, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1
    assert lib.giga_adam_step(ptr, ptr, ptr, ptr, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None) == -1      # steps count from 1
    assert lib.giga_adam_step(ptr, ptr, ptr, ptr, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == 0       # nothing to update
