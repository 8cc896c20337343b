"""Independent fp64 torch.autograd re-derivation of the rasterizer gradients.  TEST INFRASTRUCTURE ONLY.

The hand-written backward formulas restated in fgs_oracle.c (from kernels_backward.cuh / sh_utils.cuh / kernel_utils.cuh)
are checked against autograd of a *forward-only* fp64 re-implementation of the image formation model
(kernels_forward.cuh:61-160 projection + EWA, sh_utils.cuh:32-69 colour, kernels_forward.cuh:452-483 compositing).
The discrete structure (depth order, instance lists, screen bounds) is taken from the oracle's forward pass;
everything differentiable is recomputed here from the parameters. Dense [pixels x visible Gaussians] evaluation: use
small scenes only.

brute_force_forward (below) is the complementary check of that discrete structure: a render by definition that takes NOTHING from the
oracle's forward pass.
"""
from __future__ import annotations

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.48860251190291987
C2 = [1.0925484305920792, 0.94617469575755997, 0.31539156525251999, 0.54627421529603959]
C3 = [0.59004358992664352, 1.7701307697799304, 2.8906114426405538, 0.45704579946446572, 2.2852289973223288,
      1.865881662950577, 1.1195289977703462, 1.4453057213202769]


def _sh_color(sh0, sh_rest, means, cam, active):
    res = 0.5 + C0 * sh0[:, 0, :]
    if active > 1:
        d = means - cam
        d = d / d.norm(dim=1, keepdim=True)
        x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        k = sh_rest
        res = res - C1 * y * k[:, 0] + C1 * z * k[:, 1] - C1 * x * k[:, 2]
        if active > 4:
            xx, yy, zz, xy, xz, yz = x * x, y * y, z * z, x * y, x * z, y * z
            res = res + C2[0] * xy * k[:, 3] - C2[0] * yz * k[:, 4] + (C2[1] * zz - C2[2]) * k[:, 5] \
                - C2[0] * xz * k[:, 6] + C2[3] * (xx - yy) * k[:, 7]
            if active > 9:
                res = res + y * (C3[0] * yy - C3[1] * xx) * k[:, 8] + C3[2] * xy * z * k[:, 9] \
                    + y * (C3[3] - C3[4] * zz) * k[:, 10] + z * (C3[5] * zz - C3[6]) * k[:, 11] \
                    + x * (C3[3] - C3[4] * zz) * k[:, 12] + C3[7] * z * (xx - yy) * k[:, 13] \
                    + x * (C3[1] * yy - C3[0] * xx) * k[:, 14]
    return res


def autograd_reference(params: dict, settings, fwd: dict, grad_image: np.ndarray, loss_only: bool = False):
    """Returns {'image': ndarray, grads...} in fp64 (or, with loss_only, just the scalar sum(grad_image * image)). `params` holds numpy/torch arrays named as in oracle.forward,
    `settings` is an oracle.Settings, `fwd` the dict returned by oracle.forward (training mode)."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    P = {k: t(v).requires_grad_(True) for k, v in params.items()}
    means, scales, rots, opac = P['means'], P['scales'], P['rotations'], P['opacities'].reshape(-1)
    sh0, sh_rest = P['sh0'].reshape(-1, 1, 3), P['sh_rest']
    N = means.shape[0]
    sh_rest = sh_rest.reshape(N, -1, 3)
    S = settings
    W, H = S.width, S.height
    w2c = t(S.w2c)[:3, :4]
    cam = t(S.cam_position).reshape(1, 3)
    bg = t(S.bg_color).reshape(3)

    order = torch.tensor(fwd['prim_idx'].astype(np.int64))           # depth-sorted visible primitives
    V = order.numel()
    m = means[order]
    cam_pts = m @ w2c[:, :3].T + w2c[:, 3]
    depth = cam_pts[:, 2]
    x, y = cam_pts[:, 0] / depth, cam_pts[:, 1] / depth
    q = rots[order]
    qn = q / q.norm(dim=1, keepdim=True)
    r, qx, qy, qz = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    R = torch.stack([
        1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - r * qz), 2 * (qx * qz + r * qy),
        2 * (qx * qy + r * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - r * qx),
        2 * (qx * qz - r * qy), 2 * (qy * qz + r * qx), 1 - 2 * (qx * qx + qy * qy)], dim=1).reshape(V, 3, 3)
    var = torch.exp(2.0 * scales[order])
    cov3d = R @ torch.diag_embed(var) @ R.transpose(1, 2)
    clip_l, clip_r = (-0.15 * W - S.center_x) / S.focal_x, (1.15 * W - S.center_x) / S.focal_x
    clip_t, clip_b = (-0.15 * H - S.center_y) / S.focal_y, (1.15 * H - S.center_y) / S.focal_y
    xc, yc = x.clamp(clip_l, clip_r), y.clamp(clip_t, clip_b)
    j11, j22 = S.focal_x / depth, S.focal_y / depth
    zero = torch.zeros_like(depth)
    J = torch.stack([j11, zero, -j11 * xc, zero, j22, -j22 * yc], dim=1).reshape(V, 2, 3)
    JW = J @ w2c[:, :3]
    cov2d = JW @ cov3d @ JW.transpose(1, 2)
    ks = 0.1 if S.proper_antialiasing else 0.3
    a_raw, b, c_raw = cov2d[:, 0, 0], cov2d[:, 0, 1], cov2d[:, 1, 1]
    a, c = a_raw + ks, c_raw + ks
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], dim=1)
    opacity = torch.sigmoid(opac[order])
    if S.proper_antialiasing:
        det_raw = a_raw * c_raw - b * b
        opacity = opacity * torch.sqrt((det_raw / det).clamp_min(0.0)).detach()   # cfg:12 detaches the dilation term
    mean2d = torch.stack([x * S.focal_x + S.center_x, y * S.focal_y + S.center_y], dim=1)
    color = _sh_color(sh0[order], sh_rest[order], m, cam, S.active_sh_bases).clamp_min(0.0)

    # discrete structure from the oracle: membership of (tile, primitive) and screen bounds for the 8x4 sub-tile cull
    gw, gh = fwd['grid']
    T = gw * gh
    rank = torch.full((N,), -1, dtype=torch.int64)
    rank[order] = torch.arange(V)
    member = torch.zeros((T, V), dtype=torch.bool)
    member[torch.tensor(fwd['inst_keys'].astype(np.int64)), rank[torch.tensor(fwd['inst_prims'].astype(np.int64))]] = True
    sb = torch.tensor(fwd['screen_bounds'].astype(np.int64))[order]

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    tile = (ys // 12) * gw + (xs // 16)
    sx0, sy0 = (xs // 8) * 8, (ys // 4) * 4
    in_list = member[tile]                                                        # [P, V]
    overlap = (sb[None, :, 0] < (sx0 + 8)[:, None]) & (sx0[:, None] < sb[None, :, 1]) \
        & (sb[None, :, 2] < (sy0 + 4)[:, None]) & (sy0[:, None] < sb[None, :, 3])
    dx = mean2d[None, :, 0] - (xs.double() + 0.5)[:, None]
    dy = mean2d[None, :, 1] - (ys.double() + 0.5)[:, None]
    expo = -0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) - conic[None, :, 1] * dx * dy
    alpha = opacity[None, :] * torch.exp(expo.clamp_max(0.0))
    use = in_list & overlap & (alpha >= 1.0 / 255.0)
    alpha = torch.where(use, alpha, torch.zeros_like(alpha))
    # early termination: a Gaussian is blended iff the transmittance *before* it is >= 1e-4 (kf:452,477)
    T_after = torch.cumprod(1.0 - alpha, dim=1)
    T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
    alive = (T_before >= 1e-4).detach()
    # once dead, stay dead: first index where T_after < 1e-4 terminates the pixel
    alpha = torch.where(alive, alpha, torch.zeros_like(alpha))
    T_after = torch.cumprod(1.0 - alpha, dim=1)
    T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
    w = T_before * alpha
    img = w @ color + T_after[:, -1:] * bg[None, :]
    image = img.T.reshape(3, H, W)
    if loss_only:                      # the scalar <grad_image, image> in fp64: what finite differences perturb (no autograd involved)
        return float((image.detach() * t(grad_image).reshape(3, H, W)).sum())
    image.backward(t(grad_image).reshape(3, H, W))
    out = {'image': image.detach().numpy()}
    for k, v in P.items():
        out[k] = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))
    return out


def brute_force_forward(params: dict, settings, chunk: int = 2048, eps: float = 2e-5, eps_T: float = 1e-3) -> dict:
    """Image formation BY DEFINITION in fp64, using no tile, bound, bitmap, bucket, list or order of the oracle's forward pass.

    Every Gaussian that passes the primitive-level culls (depth range kf:67, opacity kf:75, degenerate quaternion kf:83, covariance
    determinant kf:144, opacity after the antialiasing factor kf:153) is evaluated at EVERY pixel centre, in depth order (the pipeline's
    key: the float32 depth, ties by index), and blended under the two per-pair rules only: alpha >= 1/255 (kf:467) and transmittance before
    the pair >= 1e-4 (kf:424,477). Everything else the pipeline does before blending -- opacity-aware screen bounds (kf:163-176), the exact
    tile test (ku:11-114), the 8x4 sub-tile test (kf:445-451), both sorts, instance lists, tile ranges, buckets -- claims to drop only pairs
    that would fail the alpha test and to keep the depth order: this is the independent check of that claim.
    Returns image [3,H,W], final_T [H,W], contributes [N] (the Gaussian is blended at >= 1 pixel) and risk [H,W]: pixels owning ANY pair
    (in a tile list of the pipeline or not) whose alpha lies within `eps` (relative) of 1/255, or whose transmittance in front of a pair lies
    within `eps_T` (relative) of 1e-4 -- where fp32 and fp64 may legitimately decide differently. Dense [pixels x Gaussians]: small scenes."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    S = settings
    W, H = S.width, S.height
    means, scales, rots = t(params['means']), t(params['scales']), t(params['rotations'])
    opac = t(params['opacities']).reshape(-1)
    N = means.shape[0]
    sh0, sh_rest = t(params['sh0']).reshape(N, 1, 3), t(params['sh_rest']).reshape(N, -1, 3)
    w2c, cam, bg = t(S.w2c)[:3, :4], t(S.cam_position).reshape(1, 3), t(S.bg_color).reshape(3)

    cam_pts = means @ w2c[:, :3].T + w2c[:, 3]
    depth = cam_pts[:, 2]
    keep = (depth >= S.near_plane) & (depth <= S.far_plane)                                   # kf:67
    opacity = torch.sigmoid(opac)
    keep &= opacity >= 1.0 / 255.0                                                            # kf:75
    norm_sq = (rots * rots).sum(dim=1)
    keep &= norm_sq >= 1e-8                                                                   # kf:83
    safe_depth = torch.where(keep, depth, torch.ones_like(depth))
    x, y = cam_pts[:, 0] / safe_depth, cam_pts[:, 1] / safe_depth
    qn = rots / norm_sq.clamp_min(1e-300).sqrt()[:, None]
    r, qx, qy, qz = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    R = torch.stack([
        1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - r * qz), 2 * (qx * qz + r * qy),
        2 * (qx * qy + r * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - r * qx),
        2 * (qx * qz - r * qy), 2 * (qy * qz + r * qx), 1 - 2 * (qx * qx + qy * qy)], dim=1).reshape(N, 3, 3)
    cov3d = R @ torch.diag_embed(torch.exp(2.0 * scales)) @ R.transpose(1, 2)
    clip_l, clip_r = (-0.15 * W - S.center_x) / S.focal_x, (1.15 * W - S.center_x) / S.focal_x
    clip_t, clip_b = (-0.15 * H - S.center_y) / S.focal_y, (1.15 * H - S.center_y) / S.focal_y
    xc, yc = x.clamp(clip_l, clip_r), y.clamp(clip_t, clip_b)
    j11, j22 = S.focal_x / safe_depth, S.focal_y / safe_depth
    zero = torch.zeros_like(depth)
    J = torch.stack([j11, zero, -j11 * xc, zero, j22, -j22 * yc], dim=1).reshape(N, 2, 3)
    JW = J @ w2c[:, :3]
    cov2d = JW @ cov3d @ JW.transpose(1, 2)
    ks = 0.1 if S.proper_antialiasing else 0.3
    a_raw, b, c_raw = cov2d[:, 0, 0], cov2d[:, 0, 1], cov2d[:, 1, 1]
    a, c = a_raw + ks, c_raw + ks
    det = a * c - b * b
    keep &= det >= 1e-6                                                                       # kf:144
    det = torch.where(keep, det, torch.ones_like(det))
    conic = torch.stack([c / det, -b / det, a / det], dim=1)
    if S.proper_antialiasing:
        opacity = opacity * torch.sqrt(((a_raw * c_raw - b * b) / det).clamp_min(0.0))
        keep &= opacity >= 1.0 / 255.0                                                        # kf:153
    mean2d = torch.stack([x * S.focal_x + S.center_x, y * S.focal_y + S.center_y], dim=1)
    color = _sh_color(sh0, sh_rest, means, cam, S.active_sh_bases).clamp_min(0.0)             # kf:430

    idx = torch.nonzero(keep).reshape(-1)
    key = depth[idx].to(torch.float32)                                                        # the pipeline's sort key (kf:204)
    idx = idx[torch.argsort(key, stable=True)]
    m2, cn, op, col = mean2d[idx], conic[idx], opacity[idx], color[idx]

    image = torch.empty((H * W, 3), dtype=torch.float64)
    final_T = torch.empty(H * W, dtype=torch.float64)
    contributes = torch.zeros(N, dtype=torch.bool)
    risk = torch.zeros(H * W, dtype=torch.bool)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    xs, ys = xs.reshape(-1).double() + 0.5, ys.reshape(-1).double() + 0.5
    for p0 in range(0, H * W, chunk):
        px, py = xs[p0:p0 + chunk, None], ys[p0:p0 + chunk, None]
        dx, dy = m2[None, :, 0] - px, m2[None, :, 1] - py
        expo = -0.5 * (cn[None, :, 0] * dx * dx + cn[None, :, 2] * dy * dy) - cn[None, :, 1] * dx * dy
        alpha = op[None, :] * torch.exp(expo.clamp_max(0.0))
        near_alpha = (alpha * 255.0 - 1.0).abs() < eps
        alpha = torch.where(alpha >= 1.0 / 255.0, alpha, torch.zeros_like(alpha))            # kf:467
        T_after = torch.cumprod(1.0 - alpha, dim=1)
        T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
        near_T = ((T_before * 1e4 - 1.0).abs() < eps_T) & (alpha > 0)
        risk[p0:p0 + chunk] = (near_alpha & (T_before >= 1e-4 * (1.0 - eps_T))).any(dim=1) | near_T.any(dim=1)
        alive = T_before >= 1e-4          # T only decreases: "done" (kf:477) <=> the transmittance in front of the pair is below the threshold
        alpha = torch.where(alive, alpha, torch.zeros_like(alpha))
        T_after = torch.cumprod(1.0 - alpha, dim=1)
        T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
        w = T_before * alpha
        T_end = T_after[:, -1] if idx.numel() else torch.ones(px.shape[0], dtype=torch.float64)
        image[p0:p0 + chunk] = (w @ col if idx.numel() else torch.zeros((px.shape[0], 3), dtype=torch.float64)) + T_end[:, None] * bg[None, :]
        final_T[p0:p0 + chunk] = T_end
        if idx.numel():
            contributes[idx[(alpha > 0).any(dim=0)]] = True
    return {'image': image.T.reshape(3, H, W).numpy(), 'final_T': final_T.reshape(H, W).numpy(), 'contributes': contributes.numpy(),
            'risk': risk.reshape(H, W).numpy(), 'order': idx.numpy()}

