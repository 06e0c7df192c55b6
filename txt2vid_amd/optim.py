"""Adam on the fused HIP kernel — drop-in for the `torch.optim.Adam(params, lr, betas)` call sites of
txt2vid/train/gan.py:93-94 (same defaults, same `state_dict()` layout: step / exp_avg / exp_avg_sq)."""
import torch

from . import functional as TF


def _densify_state(opt):
    """Every per-parameter state tensor contiguous, with the parameter's shape and dtype, in memory of its own."""
    for p, st in opt.state.items():
        for k, v in list(st.items()):
            if isinstance(v, torch.Tensor) and v.dim() > 0:
                if tuple(v.shape) != tuple(p.shape):
                    raise ValueError('optimiser state %r has shape %s for a parameter of shape %s' % (k, tuple(v.shape), tuple(p.shape)))
                dense = (v.stride() == p.stride() and v.untyped_storage().nbytes() >= (v.storage_offset() + v.numel()) * v.element_size()
                         and (p.is_contiguous() or TF.is_tap_major(p)))
                if not dense or v.dtype != p.dtype or v.device != p.device:
                    st[k] = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(v)   # the parameter's layout


def _check_state_layout(opt, p, st, keys):
    """The kernels walk parameter, gradient and state with ONE flat index through raw pointers: a state tensor assigned behind
    `load_state_dict`'s back (stride-0 expands, slices of a bigger buffer, another dtype / device) would be read out of bounds.
    Checked once per (parameter, state tensor) identity; raises instead of launching."""
    seen = opt.__dict__.setdefault('_t2v_layout_ok', {})
    sig = tuple(id(st.get(k)) for k in keys)
    if seen.get(id(p)) == sig:
        return
    for k in keys:
        v = st.get(k)
        if v is None:
            continue
        if not isinstance(v, torch.Tensor) or tuple(v.shape) != tuple(p.shape) or v.dtype != p.dtype or v.device != p.device:
            raise ValueError('optimiser state %r does not match its parameter (shape %s / %s, %s / %s, %s / %s)' % (
                k, tuple(getattr(v, 'shape', ())), tuple(p.shape), getattr(v, 'dtype', None), p.dtype, getattr(v, 'device', None), p.device))
        if v.stride() != p.stride():
            raise ValueError('optimiser state %r has strides %s for a parameter with strides %s: load it through load_state_dict '
                             '(which re-materialises foreign layouts) or assign a tensor laid out like the parameter' % (k, v.stride(), p.stride()))
        need = (v.storage_offset() + v.numel()) * v.element_size()
        if v.untyped_storage().nbytes() < need:
            raise ValueError('optimiser state %r does not own %d bytes of storage' % (k, need))
    seen[id(p)] = sig


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
        n = max(steps) if steps else 0
        if self.step_dev is not None:
            # a second capture (new batch shape): the host-side `step` entries only advance on eager / capture calls, the
            # replays in between advanced the device counter alone — carry on from whichever is further
            n = max(n, int(self.step_dev[0].item()))
            for st in self.state.values():
                if st:
                    st['step'] = n
        self.step_dev = torch.tensor([float(n), 0.0, 0.0], device=device, dtype=torch.float32)

    def load_state_dict(self, state_dict):
        """torch's loader keeps whatever strides the checkpoint's tensors had (`.to(device)` preserves them); the kernels walk the
        moments as dense arrays through raw pointers, so anything that is not a dense tensor of its own is re-materialised."""
        super().load_state_dict(state_dict)
        _densify_state(self)

    def state_dict(self):
        """Same layout as torch.optim.Adam. Under graph replay the step counter advances on the device only: fold it back
        into the per-parameter `step` entries first, so that a checkpoint resumes with the right bias corrections."""
        if self.step_dev is not None:
            n = int(self.step_dev[0].item())
            for st in self.state.values():
                if st:
                    st['step'] = n
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        TF.grad_sink_flush()                  # (no-op unless a backward pass left k-split partial sums pending)
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
                    st['exp_avg'] = TF.zeros_like(p)
                    st['exp_avg_sq'] = TF.zeros_like(p)
                _check_state_layout(self, p, st, ('exp_avg', 'exp_avg_sq'))
                st['step'] = int(st['step']) + 1
                g = p.grad
                if g.stride() != p.stride():             # the kernel walks p, g, m, v with one flat index
                    g = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(g)
                live = getattr(p, '_t2v_live_taps', None)
                if live is not None and not p.is_contiguous() and TF.is_tap_major(p):
                    # structurally dead kernel taps (ConvLSTM on a 1x1 state: gradient exactly 0, moments stay 0, update 0): only
                    # the live taps' contiguous rows are read and written
                    rows = [TF.tap_rows(t) for t in (p, g, st['exp_avg'], st['exp_avg_sq'])]
                    for t in live:
                        by_step.setdefault(st['step'], []).append(tuple(r[t] for r in rows))
                else:
                    if not (p.is_contiguous() or TF.is_tap_major(p)):
                        raise ValueError('parameter layout not supported by the fused optimiser: strides %s' % (p.stride(),))
                    by_step.setdefault(st['step'], []).append((p, g, st['exp_avg'], st['exp_avg_sq']))
                touched.append(p)
            for step, items in by_step.items():
                TF.adam_step_multi(items, group['lr'], b1, b2, group['eps'], step, self.grad_scale, self.step_dev)
        TF.bump_weight_epoch(touched)   # their packed weights are stale ...
        TF.repack_params(touched)       # ... refresh them in place right away (same addresses for graph replay)
        return None


class SGD(torch.optim.Optimizer):
    """`torch.optim.SGD(params, lr, momentum)` on the multi-tensor HIP kernel — the reference's `--sgd` branch
    (train/gan.py:86-89: momentum = beta1). Same `state_dict()` layout (momentum_buffer)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0, nesterov=False):
        if dampening != 0 or weight_decay != 0 or nesterov:
            raise NotImplementedError('the reference only passes lr and momentum')
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=0, nesterov=False))
        self.grad_scale = 1.0

    def make_capturable(self, device):
        """Nothing step-dependent lives on the host once the momentum buffers exist (they do after the eager warm-up)."""

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        _densify_state(self)

    @torch.no_grad()
    def step(self, closure=None):
        from ._lib import AdamJob
        TF.grad_sink_flush()
        touched = []
        for group in self.param_groups:
            mu = float(group['momentum'])
            first, later = [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                g = p.grad
                if g.stride() != p.stride():             # one flat index over p, g and the momentum buffer
                    g = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(g)
                if mu != 0 and 'momentum_buffer' not in st:
                    st['momentum_buffer'] = torch.empty_like(p, memory_format=torch.preserve_format)
                    first.append((p, g, st['momentum_buffer']))
                else:
                    _check_state_layout(self, p, st, ('momentum_buffer',))
                    later.append((p, g, st.get('momentum_buffer')))
                touched.append(p)
            for items, is_first in ((first, 1), (later, 0)):
                if not items:
                    continue
                arr = (AdamJob * len(items))()
                for a, (p, g, m) in zip(arr, items):
                    a.p, a.g, a.m, a.v, a.n = p.data_ptr(), g.data_ptr(), (m.data_ptr() if m is not None else None), None, p.numel()
                TF.check(TF.lib().t2v_sgd_multi(arr, len(items), group['lr'], mu, self.grad_scale, is_first, TF._stream()),
                         't2v_sgd_multi')
        TF.bump_weight_epoch(touched)
        TF.repack_params(touched)
        return None
