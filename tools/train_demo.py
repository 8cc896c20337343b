"""End-to-end training run at benchmark scale on one MI355X (synthetic data, the reference's loop order and schedules):
ground truth = the garden-like scene S1 (1 M Gaussians), rendered from 8 orbit views at 1920x1080; the model starts from a 50 %
subsample of it with perturbed means / grey colours / opacity 0.1 (Model.py:202-231-style initialisation) and trains with
0.8 L1 + 0.2 DSSIM, FusedAdam, the SH-degree schedule, adaptive density control, opacity reset and Morton re-ordering on a
compressed schedule. Prints one JSON line: PSNR before/after, Gaussian counts, iterations/s including the maintenance callbacks.

usage: python tools/train_demo.py [--n 1000000] [--iters 600] [--save-ply out.ply]   (the export feeds `bench.py --ply`: a TRAINED scene's it/s)
"""
import argparse, json, math, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import torch
from harness import densify as D
from harness import trainer as T
from harness.scenes import make_garden_like, orbit_views

ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=1_000_000); ap.add_argument('--iters', type=int, default=600)
ap.add_argument('--save-ply', default='')
a = ap.parse_args()
dev = torch.device('cuda:0')
gt_params = make_garden_like(a.n)
views = [v.to(dev) for v in orbit_views(8)]
gt = T.Gaussians(gt_params, dev)
targets = [T.render_image_benchmark(gt, v).clone() for v in views]
del gt
gen = torch.Generator().manual_seed(4)
keep = torch.randperm(a.n, generator=gen)[:a.n // 2].sort().values
init = {k: v[keep].contiguous().clone() for k, v in gt_params.items()}
init['means'] += 0.01 * torch.randn(init['means'].shape, generator=gen)
init['sh_coefficients_0'] = torch.zeros_like(init['sh_coefficients_0'])
init['sh_coefficients_rest'] = torch.zeros_like(init['sh_coefficients_rest'])
init['opacities'] = torch.full_like(init['opacities'], math.log(0.1 / 0.9))
g = T.Gaussians(init, dev, active_sh_degree=0)
g.training_setup(training_cameras_extent=5.0)
S = a.iters
schedule = dict(D.GARDEN_SCHEDULE, densification_start=S // 6, densification_end=S * 5 // 6, densification_interval=S // 12,
                opacity_reset_interval=S // 2, morton_interval=S // 3, morton_end=S * 5 // 6, sh_interval=S // 6)
psnr = lambda x, y: float(-10.0 * torch.log10(((x - y) ** 2).mean()))
mean_psnr = lambda: sum(psnr(T.render_image_benchmark(g, v), t) for v, t in zip(views, targets)) / len(views)
p0, n0 = mean_psnr(), g.means.shape[0]
counts, dgen = [], torch.Generator().manual_seed(9)
torch.cuda.synchronize(); t0 = time.perf_counter(); t_cb = 0.0
for it in range(a.iters):
    tc = time.perf_counter()
    stats = D.run_callbacks(g, it, schedule, dgen)
    if stats:
        torch.cuda.synchronize(); counts.append(stats['total']); t_cb += time.perf_counter() - tc
    v = it % len(views)
    loss = T.training_iteration(g, views[v], targets[v], it, densification_end=schedule['densification_end'])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
if a.save_ply:
    from harness.ply import save_ply
    save_ply(g, a.save_ply)
print(json.dumps({'scene': f'garden-like {a.n} Gaussians ground truth, 8 views 1920x1080', 'iterations': a.iters, 'psnr_start_db': p0, 'psnr_end_db': mean_psnr(),
                  'gaussians_start': n0, 'gaussians_end': g.means.shape[0], 'counts_after_density_control': counts, 'final_loss': float(loss),
                  'active_sh_degree': g.active_sh_degree, 'iters_per_sec_incl_callbacks': a.iters / dt,
                  'seconds_in_density_control_callbacks': t_cb, 'seconds_total': dt}))
