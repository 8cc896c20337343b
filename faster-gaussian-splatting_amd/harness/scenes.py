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


# ---- a structured ground truth for end-to-end training runs (tools/train_full.py) ------------------------------------------
def make_surface_scene(n: int, seed: int = 4321, sh_bases: int = 16, morton: bool = True, disk_scale: float = 1.3, jitter: float = 0.24) -> dict:
    """Gaussians lying ON surfaces, the way a trained 3DGS scene looks: a gently rolling textured ground over [-4, 4]^2 and a dozen
    textured ellipsoids resting on it; every Gaussian is a thin, nearly opaque disk in the tangent plane of its surface point, coloured by
    a procedural albedo (checker + stripes at 16 cm and 2.5 cm + a per-disk jitter + a per-object tint) with no view dependence. Unlike `make_garden_like` (independent random
    blobs filling a volume: every view looks like noise with parallax between the layers), renders of this scene are consistent across
    views, so a model trained on some cameras can be judged on held-out ones. World is y-down (the ground is near y = +2).
    `disk_scale` = disk sigma / disk spacing (1.3: neighbours blend into a smooth albedo; ~0.6: every disk stays a distinct cell), `jitter` = range
    of the per-disk albedo variation (large values + small disks give a mosaic whose detail a model can only reach by growing)."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    nrm = lambda *s: torch.randn(*s, generator=g)
    n_obj = 12
    centers_xz = (u(n_obj, 2) * 2.0 - 1.0) * 3.0
    radii = torch.stack([0.45 + 0.6 * u(n_obj), 0.35 + 0.8 * u(n_obj), 0.45 + 0.6 * u(n_obj)], dim=1)          # ellipsoid semi-axes (x, y, z)
    ground_y = lambda x, z: 2.0 + 0.22 * torch.sin(0.9 * x) * torch.cos(0.7 * z)
    centers = torch.stack([centers_xz[:, 0], ground_y(centers_xz[:, 0], centers_xz[:, 1]) - 0.9 * radii[:, 1], centers_xz[:, 1]], dim=1)
    areas = torch.cat([torch.tensor([64.0]), 4.0 * math.pi * ((radii[:, 0] * radii[:, 1]) ** 1.6 + (radii[:, 0] * radii[:, 2]) ** 1.6
                                                              + (radii[:, 1] * radii[:, 2]) ** 1.6).div(3.0).pow(1.0 / 1.6)])
    counts = (areas / areas.sum() * n).long()
    counts[0] += n - int(counts.sum())
    pos, nor, tint = [], [], []
    # ground
    m = int(counts[0])
    x, z = u(m) * 8.0 - 4.0, u(m) * 8.0 - 4.0
    dydx, dydz = 0.22 * 0.9 * torch.cos(0.9 * x) * torch.cos(0.7 * z), -0.22 * 0.7 * torch.sin(0.9 * x) * torch.sin(0.7 * z)
    pos.append(torch.stack([x, ground_y(x, z), z], dim=1))
    nor.append(torch.nn.functional.normalize(torch.stack([dydx, -torch.ones(m), dydz], dim=1), dim=1))        # up = -y
    tint.append(torch.tensor([0.55, 0.6, 0.45]).expand(m, 3))
    for o in range(n_obj):
        m = int(counts[1 + o])
        d = torch.nn.functional.normalize(nrm(m, 3), dim=1)
        pos.append(centers[o] + d * radii[o])
        nor.append(torch.nn.functional.normalize(d / radii[o], dim=1))
        hue = u(3) * 0.7 + 0.25
        tint.append(hue.expand(m, 3))
    pos, nor, tint = torch.cat(pos), torch.cat(nor), torch.cat(tint)
    # procedural albedo: a 0.5 m checker, 8 cm stripes and a slow gradient, modulated per object
    chk = ((torch.floor(pos[:, 0] * 2.0) + torch.floor(pos[:, 2] * 2.0) + torch.floor(pos[:, 1] * 2.0)) % 2.0) * 0.5 + 0.5
    stripes = 0.5 + 0.5 * torch.sin(pos[:, 0] * 40.0 + 3.0 * torch.sin(pos[:, 2] * 2.0))
    slow = 0.5 + 0.5 * torch.sin(pos * torch.tensor([0.8, 1.3, 1.1]) + torch.tensor([0.3, 1.1, 2.0]))
    fine = 0.5 + 0.5 * torch.sin(250.0 * (0.6 * pos[:, 0] + 0.8 * pos[:, 2]) + 60.0 * pos[:, 1])        # 2.5 cm period: ~6 pixels from the training cameras
    jitter = jitter * (u(n, 3) - 0.5)                                                                         # per-disk albedo variation (disk spacing ~7 mm)
    rgb = (tint * (0.35 + 0.65 * chk[:, None]) * (0.7 + 0.3 * stripes[:, None]) * (0.6 + 0.4 * slow) * (0.7 + 0.3 * fine[:, None]) + jitter).clamp(0.02, 0.98)
    # thin disks in the tangent plane: rotation takes the local z axis onto the normal
    spacing = math.sqrt(float(areas.sum()) / n)
    s_t = spacing * disk_scale * torch.exp(0.25 * nrm(n, 1)) * (0.8 + 0.4 * u(n, 2))
    scales = torch.log(torch.cat([s_t, 0.12 * s_t.mean(dim=1, keepdim=True)], dim=1))
    zaxis = torch.tensor([0.0, 0.0, 1.0]).expand(n, 3)
    half = torch.nn.functional.normalize(zaxis + nor + 1e-6 * nrm(n, 3), dim=1)           # quaternion of the shortest arc z -> normal: (z . h, z x h)
    w = (zaxis * half).sum(dim=1, keepdim=True)
    xyz = torch.linalg.cross(zaxis, half, dim=1)
    spin = u(n, 1) * 2.0 * math.pi                                                         # random in-plane spin (about local z), applied first
    qs = torch.cat([torch.cos(spin / 2), torch.zeros(n, 2), torch.sin(spin / 2)], dim=1)
    qa = torch.cat([w, xyz], dim=1)
    rot = torch.stack([qa[:, 0] * qs[:, 0] - (qa[:, 1:] * qs[:, 1:]).sum(dim=1),
                       qa[:, 0] * qs[:, 1] + qs[:, 0] * qa[:, 1] + qa[:, 2] * qs[:, 3] - qa[:, 3] * qs[:, 2],
                       qa[:, 0] * qs[:, 2] + qs[:, 0] * qa[:, 2] + qa[:, 3] * qs[:, 1] - qa[:, 1] * qs[:, 3],
                       qa[:, 0] * qs[:, 3] + qs[:, 0] * qa[:, 3] + qa[:, 1] * qs[:, 2] - qa[:, 2] * qs[:, 1]], dim=1)
    params = {
        'means': pos.contiguous(),
        'scales': scales.contiguous(),
        'rotations': rot.contiguous(),
        'opacities': _logit((0.75 + 0.23 * u(n, 1)).clamp(0.02, 0.98)),
        'sh_coefficients_0': rgb_to_sh0(rgb)[:, None, :].contiguous(),
        'sh_coefficients_rest': torch.zeros(n, sh_bases - 1, 3),
    }
    if morton:
        perm = morton_order(params['means'])
        params = {k: v[perm].contiguous() for k, v in params.items()}
    return params
