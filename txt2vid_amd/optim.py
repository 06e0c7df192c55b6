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
        self.step_dev = None          # device-resident step counter (set by `make_capturable`)

    def make_capturable(self, device):
        """Keep the step number on the device so that a captured HIP graph applies the right bias
        corrections on every replay (all parameters of this optimiser step together)."""
        steps = [int(self.state[p]['step']) for g in self.param_groups for p in g['params'] if p in self.state and self.state[p]]
        if len({tuple(g['betas']) for g in self.param_groups}) != 1:
            raise NotImplementedError('graph capture: one (beta1, beta2) per optimiser')
        self.step_dev = torch.tensor([float(max(steps) if steps else 0), 0.0, 0.0], device=device, dtype=torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        if self.step_dev is not None:
            b1, b2 = self.param_groups[0]['betas']
            TF.check(TF.lib().t2v_adam_tick(TF._p(self.step_dev), b1, b2, TF._stream()), 't2v_adam_tick')
        touched = []
        for group in self.param_groups:
            b1, b2 = group['betas']
            by_step = {}                      # tensors that have taken the same number of steps update in one launch
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
                by_step.setdefault(st['step'], []).append((p, g, st['exp_avg'], st['exp_avg_sq']))
                touched.append(p)
            for step, items in by_step.items():
                TF.adam_step_multi(items, group['lr'], b1, b2, group['eps'], step, self.grad_scale, self.step_dev)
        TF.bump_weight_epoch(touched)   # their packed weights are stale ...
        TF.repack_params(touched)       # ... refresh them in place right away (same addresses for graph replay)
        return None
