"""A/B of K10's tile -> workgroup mapping and of the two single-workgroup planning kernels, inside one process, S2 and the layered scene:
  mapping (fgs_debug_set_option(10, m)): 254 = device-side block plan (round 3 default), 0 = one band of tile rows per XCD (round 1/2), 1 = single
  rows interleaved; key 11 = 1 restores the rocPRIM bucket scan (and with it the bands).
Reports the stage times of blend_forward (training and inference), bucket_scan (= plan_tiles_kernel or the rocPRIM scan) and stage_pixels
(= plan_blend_backward_kernel + stage_pixels_kernel), interleaved over 4 rounds so that clock / placement drift hits every variant alike."""
import os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py'] + (['--ply', os.environ['FGS_PLY']] if os.environ.get('FGS_PLY') else [])     # FGS_PLY: a trained scene instead of S2
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
VARIANTS = (('columns-top-down (default)', 252, 0, 0), ('bands', 0, 0, 0), ('bands+rocprim-scan', 0, 1, 0), ('plan', 254, 0, 0), ('rows1', 1, 0, 0))
for shift in ((0.0,) if os.environ.get('FGS_PLY') else (0.0, -3.0)):
    p2 = dict(params); p2['opacities'] = params['opacities'] + shift
    g = T.Gaussians(p2, dev)
    g.training_setup(training_cameras_extent=5.0)
    vs = [v.to(dev) for v in views]
    S = [T.extract_settings(v, g.active_sh_bases, v.background_color) for v in vs]
    tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in vs]
    res = {}
    for rnd in range(4):
        for name, m, lib_scan, exp in VARIANTS:
            assert be.lib.fgs_debug_set_option(10, m) == 0 and be.lib.fgs_debug_set_option(11, lib_scan) == 0 and be.lib.fgs_debug_set_option(12, exp) == 0
            for i in range(2): T.training_iteration(g, vs[i], tg[i], i)
            torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
            for i in range(8): T.training_iteration(g, vs[i], tg[i], 2 + i)
            torch.cuda.synchronize(); pr = be.profile_read()
            for s in S: be.inference(*g.tensors(), s, True, True)
            torch.cuda.synchronize(); inf = be.profile_read()['blend_forward'][0] / 8; be.profile_enable(False)
            row = tuple(round(pr[k][0] / 8, 4) for k in ('blend_forward', 'bucket_scan', 'stage_pixels', 'blend_backward')) + (round(inf, 4),)
            res.setdefault(name, []).append(row)
    print(f'opacity shift {shift}: per round (blend_forward, bucket_scan / tile plan, stage_pixels incl. K11 plan, blend_backward, inference blend) ms')
    for name, v in res.items():
        best = [min(r[i] for r in v) for i in range(5)]
        print(f'   {name:28s} best {best}   rounds {v}')
be.lib.fgs_debug_set_option(10, 252); be.lib.fgs_debug_set_option(11, 0); be.lib.fgs_debug_set_option(12, 0)
