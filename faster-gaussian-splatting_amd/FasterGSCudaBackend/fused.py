"""Fused backward + Adam (the reference's `FasterGSFused` branch, README.md:37 -- not present in /root/reference).

Defined by equivalence: `render_and_step` leaves parameters and Adam state as
`diff_rasterize(...) -> loss.backward() -> FusedAdam.step()` on the reference's main branch would, without ever
materialising the 59-float per-Gaussian gradient.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch

from ._backend import RasterizerSettings, default_backend

GROUP_ORDER = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')   # Model.py:238-245


class FusedRasterizerOptimizer:
    """Holds Adam state for the six parameter tensors (in GROUP_ORDER) and performs render + backward + update."""

    def __init__(self, params: Sequence[torch.Tensor], lrs: Sequence[float], betas=(0.9, 0.999), eps: float = 1e-15) -> None:
        assert len(params) == 6 and len(lrs) == 6
        self.params = list(params)
        self.lrs = list(lrs)
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def render_and_step(self, settings: RasterizerSettings, grad_fn: Callable[[torch.Tensor], torch.Tensor],
                        densification_info: torch.Tensor | None = None) -> torch.Tensor:
        """grad_fn maps the rendered image [3,H,W] to dL/dimage (same shape). Returns the rendered image."""
        means, sh0, sh_rest, opacities, scales, rotations = self.params
        be = default_backend()
        res = be.forward(means, scales, rotations, opacities, sh0, sh_rest, settings)
        grad_image = grad_fn(res.image)
        self.step_count += 1
        be.backward_adam_fused(densification_info, grad_image, res.image, self.params, self.exp_avg, self.exp_avg_sq, res.buffers,
                               settings, res.state, self.step_count, self.lrs, self.betas, self.eps)
        return res.image
