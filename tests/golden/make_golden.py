"""Generates the committed golden fixtures from the CPU oracle (NOT from the reference, which cannot run here: no nvcc, no
NVIDIA GPU -- "parity unpinned", see oracle/fgs_oracle.c). Run from the repo root:  python tests/golden/make_golden.py

s0.npz : scene S0 (1 000 Gaussians, 128x128): inputs, every integer intermediate, image, T_final, the six
         gradients for a fixed grad_image, densification_info, and parameters/moments after 1 and 3 Adam steps.
tiny_aa.npz : 150 Gaussians, 48x36 (partial tiles), proper_antialiasing on, coloured background, SH degree 1.
"""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(REPO), str(REPO / 'faster-gaussian-splatting_amd'), str(REPO / 'tests')]
from harness.scenes import View, make_s0  # noqa: E402
from oracle import oracle as O  # noqa: E402
import helpers  # noqa: E402

KEEP = ('V', 'I', 'B', 'n_touched', 'screen_bounds', 'mean2d', 'conic_opacity', 'color', 'depth_keys', 'prim_idx', 'offsets',
        'inst_keys', 'inst_prims', 'ranges', 'n_buckets', 'bucket_offsets', 'image', 'final_T', 'n_processed', 'max_n_processed')


def dump(path, params, view, K, aa, bg, seed):
    S, _ = helpers.settings_pair(view, K, aa, bg)
    a = helpers.np_params(params)
    out = {f'in_{k}': v for k, v in zip(helpers.NAMES, a)}
    out['settings'] = np.array([K, view.width, view.height, view.focal_x, view.focal_y, view.center_x, view.center_y,
                                view.near_plane, view.far_plane, float(aa)], np.float64)
    out['w2c'], out['cam_position'], out['bg_color'] = S.w2c, S.cam_position, S.bg_color
    for bs in (32, 64):
        f = O.forward(*a, S, bucket_size=bs)
        grad_image = np.random.default_rng(seed).standard_normal(f['image'].shape).astype(np.float32)
        dens = np.zeros((2, f['N']), np.float32)
        g = O.backward(f, S, grad_image, dens)
        if bs == 32:
            out['grad_image'] = grad_image
            out.update({k: np.asarray(f[k]) for k in KEEP})
            out.update({f'grad_{k}': g[k] for k in helpers.GRAD_KEYS})
            out['densification_info'] = dens
        else:
            out['b64_bucket_offsets'], out['b64_B'] = f['bucket_offsets'], np.asarray(f['B'])
            out.update({f'b64_grad_{k}': g[k] for k in helpers.GRAD_KEYS})
    inf = O.forward(*a, S, inference=True, to_chw=False, clamp_output=True)
    out['inference_hwc_clamped'] = inf['image']
    # Adam: 3 steps on the means tensor with the oracle's gradient (adam.cu:10-71), lr 1.6e-4, eps 1e-15
    p = a[0].copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    for step in (1, 2, 3):
        O.adam_step(out['grad_means'], p, m, v, step, 1.6e-4)
        if step in (1, 3):
            out[f'adam{step}_param'], out[f'adam{step}_exp_avg'], out[f'adam{step}_exp_avg_sq'] = p.copy(), m.copy(), v.copy()
    np.savez_compressed(path, **out)
    print(path, {k: out[k] for k in ('V', 'I', 'B')}, f'{Path(path).stat().st_size / 1e6:.2f} MB')


if __name__ == '__main__':
    here = Path(__file__).resolve().parent
    params, view = make_s0()
    dump(here / 's0.npz', params, view, 16, False, None, 0)
    p2, v2 = make_s0(seed=3, n=150)
    v2 = View(v2.w2c, v2.position, 48, 36, 40.0, 40.0, 24.0, 18.0, 0.2, 1e4, torch.tensor([0.2, 0.5, 0.7]))
    dump(here / 'tiny_aa.npz', p2, v2, 4, True, None, 1)
