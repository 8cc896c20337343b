"""A from-scratch, full-schedule training run on one MI355X: the evidence that the reference's training loop (Trainer.py:86-201) runs end to
end on this backend -- random initialisation, growth by adaptive density control, opacity resets, Morton re-ordering, the SH schedule -- and
not only its kernels.

* ground truth: a STRUCTURED scene (harness.scenes.make_surface_scene, default 300 k Gaussians: textured ground + ellipsoids, every Gaussian a thin
  opaque disk on its surface, a mosaic of distinct randomly tinted cells ~2 cm across -- consistent across views, so held-out cameras mean something; `--gt-scene garden` = the benchmark's random-blob
  volume S2, whose views look like noise with parallax: a model fits its training views and does not grow, profiles/r04_train_full_garden_gt.json)
  rendered at 1920x1080 from 64 training cameras (4 rings of 16 on a hemisphere shell) and 8 HELD-OUT cameras (between the rings, other azimuths);
* model: RANDOM_INITIALIZATION of fastergs_garden.yaml -- N_POINTS = 100 000 uniform samples of the scene's bounding box, carved to the points
  inside at least one training frustum (utils.py:29-52), then Model.py:202-231 (isotropic scale = RMS distance to the 3 nearest neighbours,
  identity rotation, opacity 0.1, grey colour, SH degree 0);
* schedule: fastergs_garden.yaml UNCOMPRESSED -- 30 000 iterations, random view order (DatasetSampler(random=True), Trainer.py:84), density control
  600 -> 14 900 every 100, opacity reset every 3 000, Morton order every 5 000 until 15 000, SH degree +1 every 1 000, means lr decayed over 30 000;
  all through harness.densify.run_callbacks on the device passes of csrc/densify.hip;
* report (one JSON line): Gaussian-count curve, train / held-out PSNR at 7 000 / 15 000 / 30 000, iterations/s including the callbacks, seconds
  inside them, peak VRAM, async-forward / live-block statistics, non-finite losses.

usage: python tools/train_full.py [--gt-scene surface|garden] [--gt 300000] [--iters 30000] [--points 100000] [--max-gaussians 0] [--max-seconds 0] [--save-ply out.ply]
       --max-gaussians: density control stops ADDING above this count (0 = no cap, as the reference); --max-seconds: wall-clock guard of the loop.
"""
import argparse, json, math, os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import torch
import FasterGSCudaBackend as FGS
from harness import densify as D
from harness import trainer as T
from harness.scenes import initialize_from_point_cloud, look_at_view, make_garden_like, make_surface_scene

ap = argparse.ArgumentParser()
ap.add_argument('--gt-scene', default='surface', choices=['surface', 'garden']); ap.add_argument('--gt', type=int, default=300_000);
ap.add_argument('--disk-scale', type=float, default=0.65); ap.add_argument('--jitter', type=float, default=0.9)     # surface scene: a mosaic of distinct cells
ap.add_argument('--iters', type=int, default=30_000)
ap.add_argument('--points', type=int, default=100_000); ap.add_argument('--max-gaussians', type=int, default=0)
ap.add_argument('--max-seconds', type=float, default=0.0); ap.add_argument('--save-ply', default=''); ap.add_argument('--seed', type=int, default=7)
ap.add_argument('--eval-at', default='7000,15000,30000'); ap.add_argument('--async-forward', action='store_true')
ap.add_argument('--width', type=int, default=1920); ap.add_argument('--height', type=int, default=1080)     # smaller images: dry runs only
ap.add_argument('--schedule-scale', type=float, default=1.0, help='dry runs: every interval of the schedule times this')
ap.add_argument('--ring-size', type=int, default=16, help='training cameras per ring (4 rings; a quarter as many held-out cameras on each of 2 rings): dry runs only')
ap.add_argument('--policy', default='adc', choices=['adc', 'mcmc'], help="mcmc: the configuration's USE_MCMC variant (fastergs_garden.yaml:69,79-83,100-101; Trainer.py:120-165,199)")
ap.add_argument('--max-primitives', type=int, default=1_000_000, help='MCMC: MAX_PRIMITIVES')
a = ap.parse_args()
dev = torch.device(os.environ.get('FGS_TOOL_DEVICE', 'cuda:0'))      # the CPU dry run under tests/sim/run_with_sim.py sets this to cpu
W, H, FOCAL = a.width, a.height, 1420.0 * a.width / 1920.0


def ring(n, radius, height, phase):
    return [look_at_view((radius * math.cos(2 * math.pi * (k + phase) / n), -height, radius * math.sin(2 * math.pi * (k + phase) / n)),
                         TARGET, W, H, FOCAL) for k in range(n)]


TARGET = (0.0, 1.3, 0.0) if a.gt_scene == 'surface' else (0.0, 0.3, 0.0)
# y is down: a camera `height` above the ground plane sits at y = -height. Radii / heights keep every camera outside the scene's box.
RS, RH = a.ring_size, max(1, a.ring_size // 4)
train_views = ring(RS, 6.2, 1.0, 0.0) + ring(RS, 6.6, 2.6, 0.5) + ring(RS, 7.0, 4.2, 0.25) + ring(RS, 6.0, 0.2, 0.75)
held_views = ring(RH, 6.4, 1.8, 0.3) + ring(RH, 6.8, 3.4, 0.8)
train_views, held_views = [v.to(dev) for v in train_views], [v.to(dev) for v in held_views]

t_setup = time.perf_counter()
gt_params = make_surface_scene(a.gt, disk_scale=a.disk_scale, jitter=a.jitter) if a.gt_scene == 'surface' else make_garden_like(a.gt)
gt = T.Gaussians(gt_params, dev)
targets = [T.render_image_benchmark(gt, v).clone() for v in train_views]
held_targets = [T.render_image_benchmark(gt, v).clone() for v in held_views]
lo, hi = gt_params['means'].min(dim=0).values, gt_params['means'].max(dim=0).values
del gt, gt_params
torch.cuda.empty_cache()

# random initialisation + carving (Trainer.py:97-102, utils.py:29-52: keep the points that fall inside at least one training frustum)
gen = torch.Generator().manual_seed(a.seed)
pts = (torch.rand((a.points, 3), generator=gen) * (hi - lo) + lo).to(dev)
seen = torch.zeros(a.points, dtype=torch.bool, device=dev)
for v in train_views:
    cam = pts @ v.w2c[:3, :3].T + v.w2c[:3, 3]
    z = cam[:, 2]
    x, y = cam[:, 0] / z * v.focal_x + v.center_x, cam[:, 1] / z * v.focal_y + v.center_y
    seen |= (z > v.near_plane) & (z < v.far_plane) & (x >= 0) & (x < v.width) & (y >= 0) & (y < v.height)
pts = pts[seen].contiguous()
init = initialize_from_point_cloud(pts, use_mcmc=a.policy == 'mcmc')
g = T.Gaussians(init, dev, active_sh_degree=0)
centers = torch.stack([v.position for v in train_views])
extent = float(1.1 * (centers - centers.mean(dim=0)).norm(dim=1).max())                 # Trainer.py:91
g.training_setup(training_cameras_extent=extent)
mcmc = a.policy == 'mcmc'
if mcmc:
    D.ensure_state(g)
else:
    D.reset_densification_info(g)
FGS.set_async_forward(a.async_forward)
setup_s = time.perf_counter() - t_setup

psnr = lambda x, y: float(-10.0 * torch.log10(((x - y) ** 2).mean()))


def evaluate():
    tr = [psnr(T.render_image_benchmark(g, v), t) for v, t in zip(train_views, targets)]
    he = [psnr(T.render_image_benchmark(g, v), t) for v, t in zip(held_views, held_targets)]
    return {'train_psnr_db': sum(tr) / len(tr), 'held_out_psnr_db': sum(he) / len(he), 'held_out_min_db': min(he), 'gaussians': g.means.shape[0]}


eval_at = sorted({int(x) for x in a.eval_at.split(',') if x} | {a.iters})
schedule = dict(D.GARDEN_SCHEDULE)
if mcmc:
    schedule.update(densification_end=24_900, morton_end=25_000)
if a.schedule_scale != 1.0:
    for k in ('densification_start', 'densification_end', 'densification_interval', 'opacity_reset_interval', 'morton_interval', 'morton_end', 'sh_interval'):
        schedule[k] = max(1, int(round(schedule[k] * a.schedule_scale)))
curve, evals, events = [[0, g.means.shape[0]]], {'0': evaluate()}, []
dgen = torch.Generator().manual_seed(a.seed + 1)
order = torch.randperm(len(train_views), generator=gen).tolist()
loss_sum = torch.zeros((), device=dev)                   # summed on the device, looked at every 100 iterations (the density-control cadence)
nonfinite, t_cb, t_eval, capped_at, done = 0, 0.0, 0.0, None, 0
torch.cuda.reset_peak_memory_stats(dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
window_t, window_it, rate_curve = t0, 0, []
for it in range(a.iters):
    if a.max_gaussians and capped_at is None and g.means.shape[0] >= a.max_gaussians:
        capped_at = it
        schedule['grad_threshold'] = float('inf')        # density control keeps pruning, stops cloning / splitting (a budget guard, not in the reference)
        events.append(f'iteration {it}: {g.means.shape[0]} Gaussians >= --max-gaussians {a.max_gaussians}: growth stopped')
    tc = time.perf_counter()
    stats = D.run_mcmc_callbacks(g, it, schedule, a.max_primitives, dgen) if mcmc else D.run_callbacks(g, it, schedule, dgen)
    if stats:
        torch.cuda.synchronize(); t_cb += time.perf_counter() - tc
        curve.append([it, stats['total']])
    if it % len(order) == 0:
        order = torch.randperm(len(train_views), generator=gen).tolist()
    v = order[it % len(order)]
    if mcmc:
        loss_sum += T.training_iteration(g, train_views[v], targets[v], it, densification_end=0, before_step=lambda: D.add_mcmc_regularisation_gradients(g, 0.01, 0.01))
        D.post_optimizer_step(g, True, next(pg['lr'] for pg in g.optimizer.param_groups if pg['name'] == 'means'))
    else:
        loss_sum += T.training_iteration(g, train_views[v], targets[v], it, densification_end=schedule['densification_end'])
    done = it + 1
    if done % 100 == 0:
        s = float(loss_sum); loss_sum.zero_()
        if not math.isfinite(s):
            nonfinite += 1; events.append(f'iteration {done}: non-finite loss sum over the last 100 iterations')
    if done % 1000 == 0:
        torch.cuda.synchronize(); now = time.perf_counter()
        rate_curve.append([done, g.means.shape[0], (done - window_it) / (now - window_t)]); window_t, window_it = now, done
    if done in eval_at:
        torch.cuda.synchronize(); te = time.perf_counter()
        evals[str(done)] = evaluate()
        torch.cuda.synchronize(); t_eval += time.perf_counter() - te
    if a.max_seconds and time.perf_counter() - t0 > a.max_seconds:
        events.append(f'iteration {done}: --max-seconds {a.max_seconds} reached, loop stopped'); break
torch.cuda.synchronize(); dt = time.perf_counter() - t0 - t_eval
if str(done) not in evals:
    evals[str(done)] = evaluate()
if a.save_ply:
    from harness.ply import save_ply
    save_ply(g, a.save_ply)
final = evals[str(done)]
print(json.dumps({
    'what': 'from-scratch full-schedule training run (tools/train_full.py): random initialisation + carving, fastergs_garden.yaml schedule uncompressed'
            + (f'; USE_MCMC variant (MAX_PRIMITIVES {a.max_primitives}, densification until 24 900, noise after every step, opacity / scale regularisation 0.01)' if mcmc else ''),
    'policy': a.policy,
    'ground_truth': f'{a.gt_scene} scene (disk scale {a.disk_scale}, jitter {a.jitter}), {a.gt} Gaussians, {W}x{H}, {len(train_views)} training + {len(held_views)} held-out cameras', 'extent': extent,
    'init_points': a.points, 'gaussians_after_carving': curve[0][1], 'iterations_done': done, 'iterations_planned': a.iters,
    'gaussians_end': g.means.shape[0], 'gaussians_max': max(c[1] for c in curve), 'count_curve_every_10th_call': curve[::10] + [curve[-1]],
    'psnr': evals, 'held_out_minus_train_db': final['held_out_psnr_db'] - final['train_psnr_db'],
    'iters_per_sec_incl_callbacks': done / dt, 'seconds_total_excl_eval': dt, 'seconds_in_callbacks': t_cb, 'seconds_in_eval': t_eval, 'seconds_setup': setup_s,
    'rate_curve_iteration_gaussians_its_per_s': rate_curve, 'active_sh_degree': g.active_sh_degree,
    'peak_vram_GB': {'allocated': torch.cuda.max_memory_allocated(dev) / 1e9, 'reserved': torch.cuda.max_memory_reserved(dev) / 1e9},
    'async_forward_stats': FGS.async_forward_stats(), 'live_block_stats': FGS.live_block_stats(), 'nonfinite_loss_windows': nonfinite,
    'growth_cap': {'max_gaussians': a.max_gaussians, 'reached_at_iteration': capped_at}, 'events': events}))
