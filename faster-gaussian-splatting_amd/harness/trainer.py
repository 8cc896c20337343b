"""Stand-in for the callers of the hot path that live in the un-vendored NeRFICG framework:

* `Gaussians`        -- the six nn.Parameters + FusedAdam groups/learning rates of Model.py:51-121, 232-260
* `extract_settings` -- View -> RasterizerSettings, field for field as Renderer.py:19-43
* `training_iteration` -- the per-iteration call order of Trainer.py:170-199
Learning rates are the garden configuration's (fastergs_garden.yaml:102-110). The loss is the reference's
0.8*L1 + 0.2*DSSIM (Trainer.py:52-53) through the fused HIP kernels of harness/loss.py (`fused_dssim` itself lives in the
un-vendored NeRFICG framework; see csrc/loss.hip for what is implemented instead).
"""
from __future__ import annotations

import math

import torch

from FasterGSCudaBackend import FusedAdam, RasterizerSettings, diff_rasterize, rasterize

from .scenes import View

GARDEN_LR = {  # fastergs_garden.yaml:102-110
    'means_init': 1.6e-4, 'means_final': 1.6e-6, 'means_max_steps': 30_000,
    'sh_coefficients_0': 2.5e-3, 'sh_coefficients_rest': 1.25e-4, 'opacities': 2.5e-2, 'scales': 5.0e-3, 'rotations': 1.0e-3,
}
PARAM_ORDER = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')   # Model.py:238-245


def extract_settings(view: View, active_sh_bases: int, bg_color: torch.Tensor, proper_antialiasing: bool = False) -> RasterizerSettings:
    return RasterizerSettings(view.w2c, view.position, bg_color, active_sh_bases, view.width, view.height, view.focal_x,
                              view.focal_y, view.center_x, view.center_y, view.near_plane, view.far_plane, proper_antialiasing)


class Gaussians(torch.nn.Module):
    def __init__(self, params: dict, device, max_sh_degree: int = 3, active_sh_degree: int | None = None) -> None:
        super().__init__()
        for k in PARAM_ORDER:
            self.register_parameter(k, torch.nn.Parameter(params[k].to(device=device, dtype=torch.float32).contiguous()))
        self.max_sh_degree = max_sh_degree
        self.active_sh_degree = max_sh_degree if active_sh_degree is None else active_sh_degree
        self.densification_info = torch.zeros((2, self.means.shape[0]), dtype=torch.float32, device=device)   # Model.py:308-310
        self.optimizer: FusedAdam | None = None
        self.extent = 1.0

    @property
    def active_sh_bases(self) -> int:
        return (self.active_sh_degree + 1) ** 2

    def increase_used_sh_degree(self) -> None:                      # Model.py:144-148, every 1 000 iterations (Trainer.py:114-118)
        self.active_sh_degree = min(self.active_sh_degree + 1, self.max_sh_degree)

    def training_setup(self, training_cameras_extent: float = 1.0, lr: dict = GARDEN_LR) -> None:   # Model.py:232-253
        self.extent, self._lr = training_cameras_extent, lr
        groups = [{'params': [getattr(self, k)], 'name': k,
                   'lr': lr['means_init'] * training_cameras_extent if k == 'means' else lr[k]} for k in PARAM_ORDER]
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)

    def update_learning_rate(self, iteration: int) -> None:         # Model.py:255-260 (log-linear decay of the means lr)
        lr = self._lr
        t = min(max(iteration / lr['means_max_steps'], 0.0), 1.0)
        lr_means = math.exp(math.log(lr['means_init'] * self.extent) * (1.0 - t) + math.log(lr['means_final'] * self.extent) * t)
        for g in self.optimizer.param_groups:
            if g['name'] == 'means':
                g['lr'] = lr_means

    def tensors(self):
        return self.means, self.scales, self.rotations, self.opacities, self.sh_coefficients_0, self.sh_coefficients_rest


def render_image_training(g: Gaussians, view: View, update_densification_info: bool, bg_color: torch.Tensor) -> torch.Tensor:
    """Renderer.py:72-86."""
    return diff_rasterize(*g.tensors(),
                          densification_info=g.densification_info if update_densification_info else torch.empty(0),
                          rasterizer_settings=extract_settings(view, g.active_sh_bases, bg_color))


@torch.inference_mode()
def render_image_benchmark(g: Gaussians, view: View, to_chw: bool = True) -> torch.Tensor:
    """Renderer.py:107-123."""
    return rasterize(*g.tensors(), rasterizer_settings=extract_settings(view, g.active_sh_bases, view.background_color),
                     to_chw=to_chw, clamp_output=True)


def l1_loss(image: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return (image - target).abs().mean()


def photometric_loss(image: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """LAMBDA_L1 * L1 + LAMBDA_DSSIM * DSSIM with the garden weights 0.8 / 0.2 (fastergs_garden.yaml:98-99, Loss.py:15-16)."""
    from .loss import l1_dssim_loss
    return l1_dssim_loss(image, target, 0.8, 0.2)


_UNIT = {}


def _unit_gradient(loss: torch.Tensor) -> torch.Tensor:
    """dL/dL = 1 as a tensor that already exists: `loss.backward()` allocates and fills one per call (a 5 us kernel on the critical path of a
    2 ms iteration; profiles/r05_kernel_sequence.txt). Nothing writes to it: autograd only reads the seed."""
    key = (loss.device, loss.dtype)
    one = _UNIT.get(key)
    if one is None:
        one = _UNIT[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    return one


def training_iteration(g: Gaussians, view: View, target: torch.Tensor, iteration: int, *, densification_end: int = 14_900,
                       loss_scale: float = 1.0, before_step=None, loss_fn=photometric_loss) -> torch.Tensor:
    """One optimisation step in the reference's order (Trainer.py:170-199): lr update -> render -> loss -> backward ->
    optimizer.step -> zero_grad. `before_step` (if given) runs between backward and step (gradient exchange hook)."""
    g.update_learning_rate(iteration + 1)
    image = render_image_training(g, view, update_densification_info=iteration < densification_end, bg_color=view.background_color)
    loss = loss_fn(image, target)
    if loss_scale != 1.0:                 # (two elementwise launches per iteration otherwise)
        loss = loss * loss_scale
    loss.backward(gradient=_unit_gradient(loss))          # = loss.backward() without the fill kernel that seeds it every iteration
    if before_step is not None:
        before_step()
    g.optimizer.step()
    g.optimizer.zero_grad()
    return loss.detach()
