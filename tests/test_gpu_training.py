"""GPU end-to-end: the reference's training loop order (Trainer.py:114-199) on a small synthetic multi-view scene --
render -> 0.8 L1 + 0.2 DSSIM -> backward -> FusedAdam, with the SH-degree schedule, adaptive density control, opacity reset
and Morton re-ordering on a compressed schedule. Checks that the pieces compose: PSNR against the target views rises and
the Gaussian count changes through clone/split/prune."""
import math

import pytest
import torch

import helpers  # noqa: F401

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def test_short_training_run_with_densification(hip_backend):
    from harness import densify as D
    from harness import trainer as T
    from harness.scenes import make_s0, orbit_views
    dev = torch.device('cuda')
    gt_params, _ = make_s0(seed=21, n=1500)
    gt_params['scales'] = gt_params['scales'] + 0.3
    views = [v.to(dev) for v in orbit_views(4, radius=4.0, cam_height=0.5, width=192, height=144, focal=160.0)]
    gt = T.Gaussians(gt_params, dev)
    targets = [T.render_image_benchmark(gt, v).clone() for v in views]

    gen = torch.Generator().manual_seed(4)
    init = {k: v.clone() for k, v in gt_params.items()}
    keep = torch.randperm(1500, generator=gen)[:600]                      # start from 40 % of the Gaussians, perturbed
    init = {k: v[keep].contiguous() for k, v in init.items()}
    init['means'] += 0.05 * torch.randn(init['means'].shape, generator=gen)
    init['sh_coefficients_0'] = torch.zeros_like(init['sh_coefficients_0'])
    init['sh_coefficients_rest'] = torch.zeros_like(init['sh_coefficients_rest'])
    init['opacities'] = torch.full_like(init['opacities'], math.log(0.1 / 0.9))     # Model.py:202-231 initialisation
    g = T.Gaussians(init, dev, active_sh_degree=0)
    g.training_setup(training_cameras_extent=4.0)
    schedule = dict(D.GARDEN_SCHEDULE, densification_start=100, densification_end=350, densification_interval=50,
                    opacity_reset_interval=200, morton_interval=150, morton_end=300, sh_interval=100, grad_threshold=1e-4)

    def mean_psnr():
        return sum(_psnr(T.render_image_benchmark(g, v), t) for v, t in zip(views, targets)) / len(views)

    psnr_start, n_start = mean_psnr(), g.means.shape[0]
    counts, losses = set(), []
    dgen = torch.Generator().manual_seed(9)
    for it in range(400):
        stats = D.run_callbacks(g, it, schedule, dgen)
        if stats:
            counts.add(stats['total'])
        v = it % len(views)
        losses.append(float(T.training_iteration(g, views[v], targets[v], it, densification_end=schedule['densification_end'])))
    torch.cuda.synchronize()
    psnr_end = mean_psnr()
    assert all(math.isfinite(x) for x in losses)
    assert g.active_sh_degree == 3 and len(counts) >= 3 and g.means.shape[0] != n_start
    assert psnr_end > psnr_start + 3.0, (psnr_start, psnr_end)
    assert sum(losses[-40:]) / 40 < 0.6 * sum(losses[:40]) / 40
    for group in g.optimizer.param_groups:                                  # optimizer state followed the surgery
        p = group['params'][0]
        assert g.optimizer.state[p]['exp_avg'].shape == p.shape and p.shape[0] == g.means.shape[0]
