"""Adam on the fused HIP kernel — drop-in for the `torch.optim.Adam(params, lr, betas)` call sites of
txt2vid/train/gan.py:93-94 (same defaults, same `state_dict()` layout: step / exp_avg / exp_avg_sq)."""
import torch

from . import functional as TF


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError('the hot path uses no weight decay')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] = int(st['step']) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                TF.adam_step(p, g, st['exp_avg'], st['exp_avg_sq'], group['lr'], b1, b2, group['eps'], st['step'],
                             self.grad_scale)
        TF.bump_weight_epoch()          # packed-weight caches are now stale
        return None
