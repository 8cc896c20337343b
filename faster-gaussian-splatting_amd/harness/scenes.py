"""Synthetic scenes and views. Pure torch-CPU generation so every rank / test builds identical data.

S0      : 1 000 Gaussians, one 128x128 view           (BASELINE.json configs[0], plumbing + parity fixtures)
S1/2/3  : 1 M / 3 M / 6 M Gaussians, 1920x1080, 8 orbit views ('garden'-like statistics; configs[1..4])

The field names of View / the returned dict follow what Renderer.py:19-43 (extract_settings) reads from a NeRFICG
View and what Model.py:51-121 registers as parameters.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class View:
    """The subset of NeRFICG's View/PerspectiveCamera consumed by Renderer.py:29-43."""
    w2c: torch.Tensor          # [4,4] row-major world-to-camera, +z forward, +x right, +y down
    position: torch.Tensor     # [3] camera position in world space
    width: int
    height: int
    focal_x: float
    focal_y: float
    center_x: float
    center_y: float
    near_plane: float
    far_plane: float
    background_color: torch.Tensor  # [3]

    def to(self, device) -> 'View':
        return View(self.w2c.to(device), self.position.to(device), self.width, self.height, self.focal_x, self.focal_y,
                    self.center_x, self.center_y, self.near_plane, self.far_plane, self.background_color.to(device))


def look_at_view(eye, target, width, height, focal, near=0.2, far=1.0e4, bg=(0.0, 0.0, 0.0)) -> View:
    eye = torch.tensor(eye, dtype=torch.float64)
    target = torch.tensor(target, dtype=torch.float64)
    f = target - eye
    f = f / f.norm()
    down = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    x = torch.linalg.cross(down, f)
    x = x / x.norm()
    y = torch.linalg.cross(f, x)
    R = torch.stack([x, y, f])
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ eye
    return View(w2c.float(), eye.float(), width, height, float(focal), float(focal), width / 2.0, height / 2.0, near, far,
                torch.tensor(bg, dtype=torch.float32))


def orbit_views(n_views=8, radius=5.0, cam_height=1.5, width=1920, height=1080, focal=1420.0) -> list[View]:
    views = []
    for v in range(n_views):
        a = 2.0 * math.pi * v / n_views
        # world is y-down, so a camera 'cam_height' above the ground plane sits at y = -cam_height
        views.append(look_at_view((radius * math.cos(a), -cam_height, radius * math.sin(a)), (0.0, 0.0, 0.0), width, height, focal))
    return views


def _logit(p: torch.Tensor) -> torch.Tensor:
    return torch.log(p) - torch.log1p(-p)


def morton_order(means: torch.Tensor) -> torch.Tensor:
    """Permutation sorting points along a 30-bit Morton curve (stand-in for CudaUtils.MortonEncoding, Model.py:459-463)."""
    lo, hi = means.min(dim=0).values, means.max(dim=0).values
    q = ((means - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).to(torch.int64).clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    return torch.argsort(code, stable=True)


def make_s0(seed: int = 0, n: int = 1000, sh_bases: int = 16) -> tuple[dict, View]:
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    nrm = lambda *s: torch.randn(*s, generator=g)
    params = {
        'means': u(n, 3) * 2.0 - 1.0,
        'scales': torch.log(0.02 + 0.08 * u(n, 3)),
        'rotations': nrm(n, 4),
        'opacities': _logit(0.1 + 0.8 * u(n, 1)),
        'sh_coefficients_0': 0.5 * nrm(n, 1, 3),
        'sh_coefficients_rest': 0.1 * nrm(n, sh_bases - 1, 3),
    }
    w2c = torch.eye(4)
    w2c[2, 3] = 4.0
    view = View(w2c, torch.tensor([0.0, 0.0, -4.0]), 128, 128, 128.0, 128.0, 64.0, 64.0, 0.2, 1.0e4, torch.zeros(3))
    return params, view


def make_garden_like(n: int, seed: int = 1234, sh_bases: int = 16, morton: bool = True) -> dict:
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    nrm = lambda *s: torch.randn(*s, generator=g)
    means = u(n, 3) * torch.tensor([8.0, 4.0, 8.0]) + torch.tensor([-4.0, -1.5, -4.0])
    iso = torch.exp(math.log(0.012) + 0.6 * nrm(n, 1))
    scales = torch.log(iso * (0.5 + u(n, 3)))
    # Beta(2,2) via the median of three uniforms
    b = torch.sort(u(n, 3), dim=1).values[:, 1:2]
    params = {
        'means': means,
        'scales': scales,
        'rotations': nrm(n, 4),
        'opacities': _logit(b.clamp(0.02, 0.98)),
        'sh_coefficients_0': 0.5 * nrm(n, 1, 3),
        'sh_coefficients_rest': 0.05 * nrm(n, sh_bases - 1, 3),
    }
    if morton:
        perm = morton_order(params['means'])
        params = {k: v[perm].contiguous() for k, v in params.items()}
    return params


SCENE_SIZES = {'S1': 1_000_000, 'S2': 3_000_000, 'S3': 6_000_000}


# ---- initialisation from a point cloud (reference Model.py:202-231) -------------------------------------------------------
SH_C0 = 0.28209479177387814


def rgb_to_sh0(rgb):
    """Inverse of the degree-0 colour model colour = 0.5 + C0 * sh0 (sh_utils.cuh:32-35; the reference's utils.rgb_to_sh0)."""
    return (rgb - 0.5) / SH_C0


def root_mean_squared_knn_distances(points: torch.Tensor, k: int = 3, chunk: int = 4096) -> torch.Tensor:
    """sqrt of the mean squared distance to the k nearest other points -- the scale heuristic of 3DGS (NeRFICG's
    `compute_root_mean_squared_knn_distances`, an un-vendored dependency; restated from its use at Model.py:216).
    Brute force in chunks: O(N^2) distance evaluations on the tensor's device, fine up to a few 100 k points."""
    n = points.shape[0]
    out = torch.empty(n, dtype=points.dtype, device=points.device)
    for s in range(0, n, chunk):
        d2 = torch.cdist(points[s:s + chunk], points).square_()
        d2[torch.arange(min(chunk, n - s), device=points.device), torch.arange(s, min(s + chunk, n), device=points.device)] = float('inf')
        out[s:s + chunk] = d2.topk(min(k, n - 1), dim=1, largest=False).values.mean(dim=1).clamp_min(1e-7).sqrt()
    return out


def initialize_from_point_cloud(positions: torch.Tensor, colors: torch.Tensor | None = None, max_sh_degree: int = 3,
                                use_mcmc: bool = False) -> dict:
    """The six parameter tensors the reference builds from an SfM point cloud (Model.py:202-231): isotropic log-scale =
    RMS distance to the 3 nearest neighbours (x 0.1 for MCMC), identity rotation, opacity logit(0.1) (0.5 for MCMC),
    sh0 from the point colour (grey without colours), higher SH bands zero."""
    n = positions.shape[0]
    dev = positions.device
    sh0 = torch.full((n, 1, 3), rgb_to_sh0(0.5), dtype=torch.float32, device=dev) if colors is None else rgb_to_sh0(colors.float()[:, None, :])
    dist = root_mean_squared_knn_distances(positions.float())
    dist = dist * 0.1 if use_mcmc else dist
    rotations = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    rotations[:, 0] = 1.0
    p0 = 0.5 if use_mcmc else 0.1
    return {
        'means': positions.float().contiguous(),
        'scales': dist.log()[:, None].repeat(1, 3).contiguous(),
        'rotations': rotations,
        'opacities': torch.full((n, 1), math.log(p0 / (1.0 - p0)), dtype=torch.float32, device=dev),
        'sh_coefficients_0': sh0.contiguous(),
        'sh_coefficients_rest': torch.zeros((n, (max_sh_degree + 1) ** 2 - 1, 3), dtype=torch.float32, device=dev),
    }
