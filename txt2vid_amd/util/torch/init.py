"""Weight initialisation with the reference's semantics (txt2vid/util/torch/init.py:4-39): every Linear / Conv* / Embedding
weight is drawn by the chosen scheme (gain sqrt(2) where the owning block tagged the layer `is_residual`, layers.py:86-90), their
biases are zeroed, BatchNorm gets (1, 0). Host-side, once, on the CPU generator. The walk is children-first, left to right — the
order `nn.Module.apply` uses — because the draw ORDER decides the values for a given seed (tests/golden/init_xavier.npz)."""
import math

from torch.nn import init as schemes

DRAWN = ('Linear', 'Conv', 'Embedding')          # substrings of the class names whose weights are drawn


def _draw(kind, weight, residual):
    if not weight.is_contiguous():
        # (tap-major master copies, models/conv_lstm.py) torch fills a non-contiguous tensor in MEMORY order; the draws must land
        # in logical order like the reference's: draw into a dense temporary, then copy
        import torch
        tmp = torch.empty(weight.shape, dtype=weight.dtype, device=weight.device)
        _draw(kind, tmp, residual)
        with torch.no_grad():
            weight.copy_(tmp)
        return
    if kind == 'xavier':
        schemes.xavier_normal_(weight, gain=math.sqrt(2)) if residual else schemes.xavier_normal_(weight)
    elif kind == 'ortho':
        schemes.orthogonal_(weight, gain=math.sqrt(2)) if residual else schemes.orthogonal_(weight)
    else:                                        # 'normal': N(0, 0.02); the scheme has no gain
        schemes.normal_(weight, mean=0, std=0.02)


def _children_first(module):
    for child in module.children():
        yield from _children_first(child)
    yield module


def init(model, init_method=None):
    if init_method not in ('xavier', 'ortho', 'normal'):
        raise AssertionError('unknown init_method %r' % (init_method,))
    for layer in _children_first(model):
        cls = type(layer).__name__
        weight, bias = getattr(layer, 'weight', None), getattr(layer, 'bias', None)
        if any(tag in cls for tag in DRAWN):
            if weight is not None:
                _draw(init_method, weight, bool(getattr(layer, 'is_residual', False)))
        elif 'BatchNorm' in cls:
            if weight is not None:
                weight.data.fill_(1.0)
        else:
            continue
        if bias is not None:
            bias.data.fill_(0.0)
