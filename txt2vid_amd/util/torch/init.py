"""Weight initialisation — txt2vid/util/torch/init.py:4-39: xavier/orthogonal/normal on every
Linear/Conv/Embedding weight (gain sqrt(2) on modules tagged `is_residual`), biases 0, BatchNorm (1, 0).
Host-side, once, on the CPU generator: the draw order must match the reference for seed parity."""
import math
from functools import partial

import torch.nn.init as tinit


def _weight_init(layer, init_func=None):
    name = layer.__class__.__name__
    if 'Linear' in name or 'Conv' in name or 'Embedding' in name:
        if getattr(layer, 'weight', None) is not None:
            if getattr(layer, 'is_residual', False):
                init_func(layer.weight, gain=math.sqrt(2))
            else:
                init_func(layer.weight)
        if getattr(layer, 'bias', None) is not None:
            layer.bias.data.fill_(0.0)
    elif 'BatchNorm' in name:
        if getattr(layer, 'weight', None) is not None:
            layer.weight.data.fill_(1.0)
        if getattr(layer, 'bias', None) is not None:
            layer.bias.data.fill_(0.0)


def init(model, init_method=None):
    if init_method == 'xavier':
        f = tinit.xavier_normal_
    elif init_method == 'ortho':
        f = tinit.orthogonal_
    elif init_method == 'normal':
        f = partial(tinit.normal_, mean=0, std=0.02)
    else:
        raise AssertionError('unknown init_method %r' % (init_method,))
    model.apply(partial(_weight_init, init_func=f))
