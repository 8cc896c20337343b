"""Photometric loss of the training iteration: 0.8 * L1 + 0.2 * DSSIM (Trainer.py:52-53,190-193; Loss.py:15-16), as ONE
autograd node backed by the fused HIP kernels (csrc/loss.hip). The gradient w.r.t. the rendered image is produced together
with the loss value, so `backward` is a scalar multiply."""
from __future__ import annotations

import torch

from FasterGSCudaBackend._backend import default_backend


class _L1DSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image: torch.Tensor, target: torch.Tensor, lambda_l1: float, lambda_dssim: float) -> torch.Tensor:
        loss, grad, _ = default_backend().l1_dssim(image.contiguous(), target.contiguous(), lambda_l1, lambda_dssim, with_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_loss: torch.Tensor):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None, None


def l1_dssim_loss(image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2) -> torch.Tensor:
    return _L1DSSIM.apply(image, target, lambda_l1, lambda_dssim)
