"""PYTHONPATH=tools pytest -p pytest_poison_empty : torch.empty / empty_like / Tensor.new_empty hand out POISONED memory (NaN for floats, 0x7f.. for integers, True for
bool) -- a read of something nobody wrote then changes results instead of happening to see a fresh process' zero pages.  Dev aid, not collected by default."""
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.numel() == 0:
        return t
    if t.is_floating_point():
        t.fill_(float("nan"))
    elif t.dtype == torch.bool:
        t.fill_(True)
    elif t.dtype in (torch.int8, torch.uint8):
        t.fill_(0x7f)
    elif t.dtype == torch.int16:
        t.fill_(0x7fc0)                      # a bf16 NaN bit pattern
    else:
        t.fill_(0x7f7f7f7f if t.dtype == torch.int32 else 0x7f7f7f7f7f7f)
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


torch.empty = empty
torch.empty_like = empty_like
