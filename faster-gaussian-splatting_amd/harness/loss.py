"""Photometric loss of the training iteration: 0.8 * L1 + 0.2 * DSSIM (Trainer.py:52-53,190-193; Loss.py:15-16), as ONE
autograd node backed by the fused HIP kernels (csrc/loss.hip): forward launches the SSIM pass and keeps its derivative maps, backward
launches the gradient kernel with the upstream scalar folded in (a device read: no host sync, no `grad * upstream` pass over the image)."""
from __future__ import annotations

import torch

from FasterGSCudaBackend._backend import default_backend


class _L1DSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image: torch.Tensor, target: torch.Tensor, lambda_l1: float, lambda_dssim: float) -> torch.Tensor:
        image, target = image.contiguous(), target.contiguous()
        loss, _, scratch = default_backend().l1_dssim_forward(image, target, lambda_l1, lambda_dssim)
        ctx.save_for_backward(image, target, scratch)
        ctx.lambdas = (lambda_l1, lambda_dssim)
        return loss

    @staticmethod
    def backward(ctx, grad_loss: torch.Tensor):
        image, target, scratch = ctx.saved_tensors
        upstream = grad_loss.to(dtype=torch.float32, device=image.device)
        return default_backend().l1_dssim_backward(image, target, scratch, upstream, *ctx.lambdas), None, None, None


def l1_dssim_loss(image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2) -> torch.Tensor:
    return _L1DSSIM.apply(image, target, lambda_l1, lambda_dssim)
