"""FusedAdam: torch.optim.Adam subclass whose step() runs the HIP Adam kernel (reference torch_bindings/adam.py:6-36).

Same contract as the reference: exactly one tensor per param group, lazily created state (step / exp_avg / exp_avg_sq),
groups whose grad is None are skipped. All groups that do step are updated by ONE kernel launch (fgs_adam_step_multi).
"""
from __future__ import annotations

import torch

from ._backend import default_backend


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr, eps) -> None:
        super().__init__(params=params, lr=lr, eps=eps)

    @torch.no_grad()
    def step(self) -> None:
        batches: dict = {}
        for group in self.param_groups:
            assert len(group['params']) == 1, 'more than one tensor in group'
            param = group['params'][0]
            if param.grad is None or param.numel() == 0:
                continue
            state = self.state[param]
            if len(state) == 0:
                state['step'] = 0
                state['exp_avg'] = torch.zeros_like(param)
                state['exp_avg_sq'] = torch.zeros_like(param)
            state['step'] += 1
            key = (tuple(group['betas']), group['eps'], param.device)
            batches.setdefault(key, []).append((param.grad if param.grad.is_contiguous() else param.grad.contiguous(), param,
                                                state['exp_avg'], state['exp_avg_sq'], state['step'], group['lr']))
        for (betas, eps, _device), items in batches.items():
            for i in range(0, len(items), 8):
                g, p, m, v, s, lr = zip(*items[i:i + 8])
                default_backend().adam_step_multi(g, p, m, v, s, lr, betas[0], betas[1], eps)
