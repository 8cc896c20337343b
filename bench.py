#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FasterGS hot path (contract: see the task statement / DESIGN.md 'Measurement').

One "step" = one full training iteration of the reference (Trainer.py:170-199): lr update -> diff_rasterize forward ->
0.8*L1 + 0.2*DSSIM loss -> backward -> FusedAdam.step -> zero_grad, on one 1920x1080 view of the synthetic garden-like scene
(scene S2 = 3 M Gaussians by default). With N GPUs every rank renders a different orbit view per step (weak
scaling). Default exchange (--dp-mode sharded, harness/sharded.py): every rank OWNS N/G Gaussians, projects them for the G views
and ships 56-byte projected records to the renderers, which return 36-byte pixel-space gradient accumulators; K12 + Adam run
on the owner's shard. --dp-mode zero1|allreduce: replicated parameters, the 236-B/Gaussian gradient crosses xGMI instead.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene S1|S2|S3] [--no-cpu-baseline] [--no-extras]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path[:0] = [str(REPO), str(REPO / 'faster-gaussian-splatting_amd')]

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scene', default='S2', choices=['S0', 'S1', 'S2', 'S3'])
    ap.add_argument('--n-gaussians', type=int, default=0, help='override the scene size (debug)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the inference / fused side measurements')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='time only the oracle (no GPU needed)')
    ap.add_argument('--dp-mode', default='sharded', choices=['sharded', 'zero1', 'allreduce'],
                    help="N > 1: 'sharded' = every rank owns N/G Gaussians, 56-B records / 36-B accumulators cross xGMI (harness/sharded.py); "
                         "'zero1' / 'allreduce' = replicated parameters, 236-B gradients cross xGMI (harness/distributed.py)")
    return ap.parse_args()


def build_scene(args):
    from harness.scenes import SCENE_SIZES, make_garden_like, make_s0, orbit_views
    if args.scene == 'S0':
        params, view = make_s0()
        return params, [view], 'S0: 1k Gaussians, 128x128'
    n = args.n_gaussians or SCENE_SIZES[args.scene]
    params = make_garden_like(n)
    return params, orbit_views(8), f'{args.scene}: {n} garden-like Gaussians (SH degree 3), 1920x1080, 8 orbit views'


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summaries (profiles/r01_s2_pmc_*.txt, separate
    FETCH_SIZE / WRITE_SIZE passes of this same command on S2; counters are in KiB, and FETCH_SIZE reports half of the wide
    reads on gfx950 -- calibrated on the Adam kernel). Counters cannot be collected from inside the timed run."""
    try:
        vals = {}
        for key in ('fetch', 'write'):
            for line in (REPO / 'profiles' / f'r01_s2_pmc_{key}_size.txt').read_text().splitlines():
                if kernel.split('<')[0] in line and f'{key.upper()}_SIZE' in line:
                    vals[key] = float(line.split()[-1]) * 1024.0
                    break
        return 2.0 * vals['fetch'] + vals['write'], 'profiles/r01_s2_pmc_{fetch,write}_size.txt: 2 x FETCH_SIZE + WRITE_SIZE per launch (bytes)'
    except Exception as exc:
        return None, f'PMC summary not readable: {exc}'


def cpu_baseline(params, views, stats: dict, budget_s: float = 12.0) -> dict:
    """Times full training iterations (forward + backward + Adam on all 59 floats per Gaussian, one view each) of the same
    workload on the host cores with the CPU oracle (a port of the reference arithmetic, oracle/fgs_oracle.c; OpenMP over
    Gaussians / tiles / buckets): as many of the orbit views as fit a ~12 s budget. Reported baseline, not a target."""
    from oracle import oracle as O
    names = ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')
    groups = (('means', 'means', 1.6e-4), ('sh_coefficients_0', 'sh0', 2.5e-3), ('sh_coefficients_rest', 'sh_rest', 1.25e-4),
              ('opacities', 'opacities', 2.5e-2), ('scales', 'scales', 5e-3), ('rotations', 'rotations', 1e-3))
    P = {k: np.ascontiguousarray(params[k].numpy().copy()) for k in names}            # state lives outside the timed region
    M = {k: np.zeros_like(P[k]) for k in names}
    V = {k: np.zeros_like(P[k]) for k in names}
    t_f = t_b = t_a = 0.0
    done, last = 0, None
    for it, view in enumerate(views):
        S = O.Settings(view.w2c.numpy(), view.position.numpy(), view.background_color.numpy(), 16, view.width, view.height,
                       view.focal_x, view.focal_y, view.center_x, view.center_y, view.near_plane, view.far_plane, False)
        t0 = time.perf_counter()
        f = O.forward(*[P[k] for k in names], S, bucket_size=32)
        t1 = time.perf_counter()
        gi = np.sign(f['image']).astype(np.float32) / f['image'].size
        dens = np.zeros((2, f['N']), np.float32)
        g = O.backward(f, S, gi, dens)
        t2 = time.perf_counter()
        for k, gk, lr in groups:
            O.adam_step(np.ascontiguousarray(g[gk].reshape(P[k].shape)), P[k], M[k], V[k], it + 1, lr)
        t3 = time.perf_counter()
        t_f, t_b, t_a, done, last = t_f + t1 - t0, t_b + t2 - t1, t_a + t3 - t2, done + 1, f
        if t_f + t_b + t_a >= budget_s:
            break
    total = t_f + t_b + t_a
    view = views[0]
    return {'value': done / total, 'unit': 'iters/s', 'cores': O.num_threads(), 'kind': 'port',
            'sample': f'{done} full training iterations (one orbit view each; per iteration fwd {t_f / done:.2f}s + bwd {t_b / done:.2f}s + '
                      f'Adam {t_a / done:.2f}s) of the same workload, last view V={last["V"]} I={last["I"]} B32={last["B"]}; '
                      f'OpenMP threads = cores; {total:.1f} s of CPU work',
            'render_mpix_per_s': view.width * view.height / 1e6 / (t_f / done)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and not args.cpu_baseline_only:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}')
    params, views, workload = build_scene(args)

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(params, views, {})))
        return

    import torch.distributed as dist
    from FasterGSCudaBackend import FusedRasterizerOptimizer
    from FasterGSCudaBackend._backend import default_backend
    from harness import trainer as T

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (the HIP path has no CPU fallback)')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):
        dist.init_process_group('nccl', device_id=device)
    be = default_backend()
    if 'FGS_BACKWARD_VARIANT' in os.environ:      # A/B switch of the blend-backward formulation (debug)
        be.lib.fgs_debug_set_backward_variant(int(os.environ['FGS_BACKWARD_VARIANT']))

    g = T.Gaussians(params, device)
    g.training_setup(training_cameras_extent=5.0)
    n = g.means.shape[0]
    views = [v.to(device) for v in views]
    my_views = [views[(i * world + rank) % len(views)] for i in range(len(views))]

    # fixed targets: renders of a perturbed copy of the scene, and realised V / I / B per view
    stats = {}
    targets = {}
    with torch.no_grad():
        gen = torch.Generator(device='cpu').manual_seed(99)
        pert = [t.detach().clone() for t in g.tensors()]
        pert[4] = pert[4] + 0.15 * torch.randn(pert[4].shape, generator=gen).to(device)
        pert[0] = pert[0] + 0.002 * torch.randn(pert[0].shape, generator=gen).to(device)
        for v in {id(v): v for v in my_views}.values():
            S = T.extract_settings(v, g.active_sh_bases, v.background_color)
            targets[id(v)] = be.inference(*pert, S, True, True)
            res = be.forward(*g.tensors(), S)
            lay = be.blob_layout(1, n, v.width, v.height, res.state[1], res.state[2])
            n_tiles = ((v.width + 15) // 16) * ((v.height + 11) // 12)
            b_real = int(be.view(res.buffers[1], lay, 'bucket_offsets', torch.int32)[n_tiles - 1].item())
            stats[id(v)] = {'V': res.state[0], 'I': res.state[1], 'B': b_real}
            del res
        del pert

    # N > 1 (or any torch.distributed.run launch): view-parallel step of harness/distributed.py -- parameters and gradients live
    # in ONE contiguous arena each, so a step costs one reduce-scatter + one all-gather (zero1: Adam on 1/N of the arena per
    # rank) or one all-reduce. The rasterizer is called through the backend directly (no autograd copies of the 708 MB arena).
    launched_distributed = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    vp = None
    sharded = launched_distributed and args.dp_mode == 'sharded'
    if launched_distributed:
        lr = T.GARDEN_LR
        lrs = {'means': lr['means_init'] * 5.0, **{k: lr[k] for k in T.PARAM_ORDER[1:]}}
        full = {k: getattr(g, k).detach() for k in T.PARAM_ORDER}
        if sharded:
            # Gaussian-sharded step: rank r owns Gaussians r::G; only projected records and pixel-space accumulators cross xGMI
            from harness.sharded import ShardedTrainer, shard_of
            vp = ShardedTrainer(be, shard_of(full, rank, world), lrs)
        else:
            from harness.distributed import ViewParallelTrainer
            vp = ViewParallelTrainer(be, full, lrs, mode=args.dp_mode)
    settings_of = {id(v): T.extract_settings(v, g.active_sh_bases, v.background_color) for v in views}

    def step(i: int) -> None:
        v = my_views[i % len(my_views)]
        if vp is None:
            T.training_iteration(g, v, targets[id(v)], i)
        elif sharded:       # every rank names the same global batch: view (i*G + r) is rendered by rank r
            batch = [views[((i % len(my_views)) * world + r) % len(views)] for r in range(world)]
            vp.step([settings_of[id(b)] for b in batch], targets[id(v)])
        else:
            vp.step(settings_of[id(v)], targets[id(v)])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    fence()
    be.profile_enable(True)
    be.profile_read()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    prof = be.profile_read()
    be.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    used = [my_views[(args.warmup + i) % len(my_views)] for i in range(args.steps)]
    mean = lambda key: float(np.mean([stats[id(v)][key] for v in used]))
    V, I, B = mean('V'), mean('I'), mean('B')
    W_, H_ = views[0].width, views[0].height
    P_, T_ = W_ * H_, ((W_ + 15) // 16) * ((H_ + 11) // 12)
    K_ = g.active_sh_bases
    # algorithmic bytes per stage (DESIGN.md 'Measurement'). N Gaussians, V visible, I instances,
    # B buckets(64), P pixels, T tiles, K active SH bases -- all realised values of the timed views.
    stage_bytes = {
        'preprocess': 48.0 * n + (12 * K_ + 56.0) * V,
        'depth_sort': 68.0 * V,
        'offsets_scan': 20.0 * V,
        'create_instances': 40.0 * V + 6.0 * I,
        'tile_sort': 26.0 * I,
        'extract_ranges': 2.0 * I + 8.0 * T_,
        'bucket_scan': 12.0 * T_,
        'blend_forward': 48.0 * I + 8.0 * T_ + 20.0 * P_ + 3076.0 * B,
        'stage_pixels': 32.0 * P_,
        'blend_backward': 76.0 * I + 3076.0 * B,
        'preprocess_backward': 4.0 * n + 128.0 * V + 56.0 * n,           # + every one of the 14 small gradients written once
        'sh_rest_backward': 24.0 * K_ * V + 12.0 * (K_ - 1) * n,         # + the [N,K-1,3] gradient written once
        'adam': 1652.0 * n / (world if (vp is not None and args.dp_mode != 'allreduce') else 1),     # zero1 / sharded: Adam on 1/G
        'l1_dssim_loss': (24.0 + 36.0 + 48.0) * P_,       # fwd: x,y in + 3 maps out; bwd: 3 maps + x,y in, grad out (3 channels)
    }
    kernel_of = {'preprocess': 'preprocess_kernel<false>', 'blend_backward': 'blend_backward_kernel', 'adam': 'adam_kernel',
                 'blend_forward': 'blend_kernel<true>', 'create_instances': 'create_instances_kernel<u16>',
                 'tile_sort': 'rocprim radix_sort_onesweep (u16 keys)', 'sh_rest_backward': 'sh_rest_backward_kernel<false>',
                 'preprocess_backward': 'preprocess_backward_kernel<false>', 'depth_sort': 'rocprim radix_sort_onesweep (u32 keys)'}
    per_launch = {k: (v_[0] / max(v_[1], 1)) * (v_[1] / args.steps) for k, v_ in prof.items() if v_[1] > 0}   # ms per step
    dom = max((k for k in per_launch if k in stage_bytes), key=per_launch.get)
    dom_s = per_launch[dom] * 1e-3
    achieved = stage_bytes[dom] / dom_s / 1e9 if dom_s > 0 else 0.0
    bytes_iter = 1960.0 * n + (36 * K_ + 312.0) * V + 158.0 * I + 6152.0 * B + 52.0 * P_ + 28.0 * T_
    traffic, traffic_note = pmc_traffic(kernel_of.get(dom, dom)) if args.scene == 'S2' and not args.n_gaussians else (None, 'no PMC summary for this scene')
    out = {
        'metric': 'train_iters_per_sec', 'value': args.steps * world / elapsed, 'unit': 'iters/s (1 view each, whole job)',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': workload + '; full training iteration fwd+loss+bwd+Adam (BASELINE.json configs[2]), loss 0.8*L1+0.2*DSSIM, '
                               'densification_info updated', 'parallelism': f'view-parallel dp{world} ({args.dp_mode})' if vp is not None else 'single GPU',
                   'n_gaussians': n, 'visible': V, 'instances': I, 'buckets64': B, 'active_sh_bases': K_},
        'roofline': {'bound': 'hbm', 'kernel': kernel_of.get(dom, dom), 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_note, 'avg_kernel_ms': dom_s * 1e3,
                     'algorithmic_bytes_per_launch': stage_bytes[dom],
                     'note': 'dominant = longest kernel of the timed region (HIP events on the launch stream); traffic: see profiles/ PMC summaries',
                     'iteration_algorithmic_GB': bytes_iter / 1e9,
                     'iteration_frac_of_hbm_peak': bytes_iter / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        'stage_ms_per_step': {k: v[0] / args.steps for k, v in prof.items() if v[1] > 0},
        'stage_algorithmic_GBps': {k: stage_bytes[k] / (per_launch[k] * 1e-3) / 1e9 for k in per_launch if k in stage_bytes},
    }

    if rank == 0 and not args.no_extras:
        # BASELINE.json configs[1]: forward render only (the reference's render_image_benchmark path)
        v = my_views[0]
        for _ in range(3):
            T.render_image_benchmark(g, v)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            T.render_image_benchmark(g, v)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / reps
        out['render_mpix_per_sec'] = P_ / 1e6 / dt
        out['render_ms_per_frame'] = dt * 1e3
        # BASELINE.json configs[3]: fused backward + Adam
        fo = FusedRasterizerOptimizer([getattr(g, k).detach() for k in T.PARAM_ORDER],
                                      [1.6e-4 * 5.0, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3])
        tgt = targets[id(v)]
        S = T.extract_settings(v, g.active_sh_bases, v.background_color)
        grad_fn = lambda img: be.l1_dssim(img, tgt, 0.8, 0.2)[1]
        for _ in range(2):
            fo.render_and_step(S, grad_fn, g.densification_info)
        torch.cuda.synchronize(device)
        be.profile_enable(True)
        be.profile_read()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            fo.render_and_step(S, grad_fn, g.densification_info)
        torch.cuda.synchronize(device)
        out['fused_train_iters_per_sec'] = reps / (time.perf_counter() - t0)
        out['fused_stage_ms_per_step'] = {k: v_[0] / reps for k, v_ in be.profile_read().items() if v_[1] > 0}
        be.profile_enable(False)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline(params, [v.to('cpu') for v in views], stats)
        except Exception as exc:   # the baseline must never take the GPU number down with it
            out['cpu_baseline'] = {'value': None, 'unit': 'iters/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {exc}'}
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
