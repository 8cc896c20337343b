"""FusedAdam for the HIP backend.

Public contract = the reference's optimizer (torch_bindings/adam.py:6-36): a `torch.optim.Adam` subclass constructed as
`FusedAdam(param_groups, lr, eps)`, ONE tensor per parameter group, moments created on first use, groups without a gradient
left untouched, `state[param]` holding `step` (int) / `exp_avg` / `exp_avg_sq` so that NeRFICG-style optimizer surgery
(extend / prune / sort of the moments) keeps working.

What differs is the execution: the reference issues one `adam_step` launch per group (6 per iteration); here every group that
has a gradient is collected first and the whole optimizer step is ONE `fgs_adam_step_multi` launch (up to 8 groups per launch).
"""
from __future__ import annotations

from typing import Iterator, NamedTuple

import torch

from ._backend import default_backend
from .rasterization import clear_live_blocks, match_live_blocks, take_async_overflow

_GROUPS_PER_LAUNCH = 8          # AdamArgs::g[8] in csrc/fgs_kernels.h


class _Update(NamedTuple):
    grad: torch.Tensor
    param: torch.Tensor
    exp_avg: torch.Tensor
    exp_avg_sq: torch.Tensor
    step: int
    lr: float


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr, eps) -> None:
        super().__init__(params=params, lr=lr, eps=eps)

    def _moments(self, tensor: torch.Tensor) -> dict:
        """Optimizer state of `tensor`, zero-initialised the first time it is stepped (adam.py:17-21 of the reference)."""
        entry = self.state[tensor]
        if not entry:
            entry.update(step=0, exp_avg=torch.zeros_like(tensor), exp_avg_sq=torch.zeros_like(tensor))
        return entry

    def _pending(self) -> Iterator[tuple[tuple, _Update]]:
        """(launch key, update) for every group that has something to do this step."""
        for group in self.param_groups:
            tensors = group['params']
            if len(tensors) != 1:
                raise ValueError(f'FusedAdam expects one tensor per parameter group, group {group.get("name", "?")} has {len(tensors)}')
            tensor = tensors[0]
            gradient = tensor.grad
            if gradient is None or tensor.numel() == 0:
                continue
            moments = self._moments(tensor)
            moments['step'] += 1
            beta1, beta2 = group['betas']
            yield (float(beta1), float(beta2), float(group['eps']), tensor.device), _Update(
                gradient.contiguous(), tensor, moments['exp_avg'], moments['exp_avg_sq'], moments['step'], float(group['lr']))

    @torch.no_grad()
    def step(self) -> None:
        owned = [p for group in self.param_groups for p in group['params']]
        if take_async_overflow(owned):
            # the rasterizer's backward pass returned zeros because its (asynchronously sized) forward pass was truncated: stepping on them would
            # decay the moments, advance the step counts and move every parameter on momentum for a loss that was never evaluated. The mark names
            # the parameters of that pass: only the optimizer that owns them skips (and consumes the mark), any other FusedAdam steps normally
            clear_live_blocks(owned)
            return
        launches: dict[tuple, list[_Update]] = {}
        for key, update in self._pending():
            launches.setdefault(key, []).append(update)
        backend = default_backend()
        for (beta1, beta2, eps, _device), updates in launches.items():
            for first in range(0, len(updates), _GROUPS_PER_LAUNCH):
                chunk = updates[first:first + _GROUPS_PER_LAUNCH]
                # gradients that are still exactly what the rasterizer's backward pass wrote come with its per-block "any visible" flags: the
                # zeros of dead blocks are not read back (rasterization.match_live_blocks; bit-identical result, ~4 % less optimizer traffic)
                live = match_live_blocks([u.grad for u in chunk], owned) if len(updates) <= _GROUPS_PER_LAUNCH else None
                backend.adam_step_multi([u.grad for u in chunk], [u.param for u in chunk], [u.exp_avg for u in chunk],
                                        [u.exp_avg_sq for u in chunk], [u.step for u in chunk], [u.lr for u in chunk], beta1, beta2, eps,
                                        live_blocks=live)
        clear_live_blocks(owned)
