"""`FasterGSCudaBackend._C` -- the eight entry points the reference's pybind11 module exports (torch_bindings/bindings.cpp:12-21),
with the same names, positional argument order, return values and in-place behaviour, so that the reference's own
torch_bindings/*.py (rasterization.py:43-110, adam.py:27-36, filter3d.py:21-35, densification.py:11-22) run unmodified on top of it:

    forward(means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, w2c, cam_position, bg_color,
            active_sh_bases, width, height, focal_x, focal_y, center_x, center_y, near_plane, far_plane, proper_antialiasing)
        -> (image, primitive_buffers, tile_buffers, instance_buffers, bucket_buffers, n_instances, n_buckets, selector)      rasterization_api.h:8-28
    backward(densification_info, grad_image, image, means, scales, rotations, opacities, sh_coefficients_rest,
             primitive_buffers, tile_buffers, instance_buffers, bucket_buffers, w2c, cam_position, bg_color, <10 scalars>,
             n_instances, n_buckets, selector) -> six gradients                                                              rasterization_api.h:30-59
    inference(<forward arguments>, to_chw, clamp_output) -> image                                                            rasterization_api.h:61-83
    pruning_scores(scores, <forward arguments>) -> None (accumulates into scores)                                            rasterization_api.h:85-106
    adam_step(param_grad, param, exp_avg, exp_avg_sq, step_count, learning_rate, beta1, beta2, epsilon) -> None             adam/include/adam.h:7-16
    update_3d_filter(positions, w2c, filter_3d, visibility_mask, width, height, focal_x, focal_y, center_x, center_y,
                     near_plane, clipping_tolerance, distance2filter) -> None                                               filter3d/include/filter3d.h:7-20
    relocation_adjustment(old_opacities, old_scales, n_samples_per_primitive) -> (new_opacities, new_scales)                densification_api.h:8-12
    add_noise(raw_scales, raw_rotations, raw_opacities, random_samples, means, current_lr) -> None                          densification_api.h:14-21

The four scratch tensors are opaque uint8 blobs exactly as in the reference (utils/torch_utils.h:6-12: grown through a resize
callback, handed back to `backward` untouched); the three integers are the reference's (n_instances, n_buckets, selector). The
number of visible primitives -- a fourth piece of state this backend's sharded multi-GPU path uses -- stays in the primitive blob
on the device, so nothing is smuggled through the integers.
"""
from __future__ import annotations

import torch

from ._backend import RasterizerSettings
from ._backend import default_backend as _backend


def _settings(w2c, cam_position, bg_color, active_sh_bases, width, height, focal_x, focal_y, center_x, center_y, near_plane, far_plane,
              proper_antialiasing) -> RasterizerSettings:
    return RasterizerSettings(w2c, cam_position, bg_color, int(active_sh_bases), int(width), int(height), float(focal_x), float(focal_y),
                              float(center_x), float(center_y), float(near_plane), float(far_plane), bool(proper_antialiasing))


def forward(means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, opacities: torch.Tensor, sh_coefficients_0: torch.Tensor,
            sh_coefficients_rest: torch.Tensor, w2c: torch.Tensor, cam_position: torch.Tensor, bg_color: torch.Tensor, active_sh_bases: int,
            width: int, height: int, focal_x: float, focal_y: float, center_x: float, center_y: float, near_plane: float, far_plane: float,
            proper_antialiasing: bool):
    S = _settings(w2c, cam_position, bg_color, active_sh_bases, width, height, focal_x, focal_y, center_x, center_y, near_plane, far_plane,
                  proper_antialiasing)
    res = _backend().forward(means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, S)
    _n_visible, n_instances, n_buckets, selector = res.state
    return (res.image, *res.buffers, n_instances, n_buckets, selector)


def backward(densification_info: torch.Tensor, grad_image: torch.Tensor, image: torch.Tensor, means: torch.Tensor, scales: torch.Tensor,
             rotations: torch.Tensor, opacities: torch.Tensor, sh_coefficients_rest: torch.Tensor, primitive_buffers: torch.Tensor,
             tile_buffers: torch.Tensor, instance_buffers: torch.Tensor, bucket_buffers: torch.Tensor, w2c: torch.Tensor,
             cam_position: torch.Tensor, bg_color: torch.Tensor, active_sh_bases: int, width: int, height: int, focal_x: float, focal_y: float,
             center_x: float, center_y: float, near_plane: float, far_plane: float, proper_antialiasing: bool, n_instances: int,
             n_buckets: int, instance_primitive_indices_selector: int):
    S = _settings(w2c, cam_position, bg_color, active_sh_bases, width, height, focal_x, focal_y, center_x, center_y, near_plane, far_plane,
                  proper_antialiasing)
    # n_visible (state[0]) is not needed to re-derive the buffer layout (backward.cu:46-52 replays it from N, n_tiles, n_instances, n_buckets)
    state = (0, int(n_instances), int(n_buckets), int(instance_primitive_indices_selector))
    return _backend().backward(densification_info, grad_image, image, means, scales, rotations, opacities, sh_coefficients_rest,
                                      (primitive_buffers, tile_buffers, instance_buffers, bucket_buffers), S, state)


def inference(means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, opacities: torch.Tensor, sh_coefficients_0: torch.Tensor,
              sh_coefficients_rest: torch.Tensor, w2c: torch.Tensor, cam_position: torch.Tensor, bg_color: torch.Tensor, active_sh_bases: int,
              width: int, height: int, focal_x: float, focal_y: float, center_x: float, center_y: float, near_plane: float, far_plane: float,
              proper_antialiasing: bool, to_chw: bool, clamp_output: bool) -> torch.Tensor:
    S = _settings(w2c, cam_position, bg_color, active_sh_bases, width, height, focal_x, focal_y, center_x, center_y, near_plane, far_plane,
                  proper_antialiasing)
    return _backend().inference(means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, S, bool(to_chw),
                                       bool(clamp_output))


def pruning_scores(scores: torch.Tensor, means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, opacities: torch.Tensor,
                   sh_coefficients_0: torch.Tensor, sh_coefficients_rest: torch.Tensor, w2c: torch.Tensor, cam_position: torch.Tensor,
                   bg_color: torch.Tensor, active_sh_bases: int, width: int, height: int, focal_x: float, focal_y: float, center_x: float,
                   center_y: float, near_plane: float, far_plane: float, proper_antialiasing: bool) -> None:
    S = _settings(w2c, cam_position, bg_color, active_sh_bases, width, height, focal_x, focal_y, center_x, center_y, near_plane, far_plane,
                  proper_antialiasing)
    _backend().pruning_scores(scores, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, S)


def adam_step(param_grad: torch.Tensor, param: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step_count: int,
              learning_rate: float, beta1: float, beta2: float, epsilon: float) -> None:
    _backend().adam_step(param_grad, param, exp_avg, exp_avg_sq, int(step_count), float(learning_rate), float(beta1), float(beta2),
                                float(epsilon))


def update_3d_filter(positions: torch.Tensor, w2c: torch.Tensor, filter_3d: torch.Tensor, visibility_mask: torch.Tensor, width: int,
                     height: int, focal_x: float, focal_y: float, center_x: float, center_y: float, near_plane: float,
                     clipping_tolerance: float, distance2filter: float) -> None:
    _backend().update_3d_filter(positions, w2c, filter_3d, visibility_mask, width, height, focal_x, focal_y, center_x, center_y,
                                       near_plane, clipping_tolerance, distance2filter)


def relocation_adjustment(old_opacities: torch.Tensor, old_scales: torch.Tensor, n_samples_per_primitive: torch.Tensor):
    return _backend().relocation_adjustment(old_opacities, old_scales, n_samples_per_primitive)


def add_noise(raw_scales: torch.Tensor, raw_rotations: torch.Tensor, raw_opacities: torch.Tensor, random_samples: torch.Tensor,
              means: torch.Tensor, current_lr: float) -> None:
    _backend().add_noise(raw_scales, raw_rotations, raw_opacities, random_samples, means, float(current_lr))
