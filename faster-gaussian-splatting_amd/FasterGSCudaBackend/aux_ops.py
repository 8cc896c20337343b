"""update_3d_filter / relocation_adjustment / add_noise with the reference's signatures (torch_bindings/filter3d.py:6-35,
densification.py:6-22). Outside the garden configuration's hot path (FILTER_3D.USE / USE_MCMC are false there)."""
from __future__ import annotations

import torch

from ._backend import default_backend


def _gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError('FasterGS backend: tensors must live on a ROCm/HIP device (no CPU implementation)')


def update_3d_filter(positions: torch.Tensor, w2c: torch.Tensor, filter_3d: torch.Tensor, visibility_mask: torch.Tensor, width: int,
                     height: int, focal_x: float, focal_y: float, center_x: float, center_y: float, near_plane: float,
                     clipping_tolerance: float, distance2filter: float) -> None:
    _gpu(positions)
    default_backend().update_3d_filter(positions, w2c, filter_3d, visibility_mask, width, height, focal_x, focal_y, center_x, center_y,
                                       near_plane, clipping_tolerance, distance2filter)


def relocation_adjustment(old_opacities: torch.Tensor, old_scales: torch.Tensor,
                          n_samples_per_primitive: torch.Tensor) -> 'tuple[torch.Tensor, torch.Tensor]':
    _gpu(old_opacities)
    return default_backend().relocation_adjustment(old_opacities, old_scales, n_samples_per_primitive)


def add_noise(raw_scales: torch.Tensor, raw_rotations: torch.Tensor, raw_opacities: torch.Tensor, means: torch.Tensor,
              current_lr: float) -> None:
    _gpu(means)
    random_samples = torch.randn_like(means)          # densification.py:20 draws the noise on the caller side as well
    default_backend().add_noise(raw_scales, raw_rotations, raw_opacities, random_samples, means, current_lr)
