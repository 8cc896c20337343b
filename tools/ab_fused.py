"""A/B of fgs_backward_adam_fused inside ONE process: single kernel (option 3 = 1) vs round 1's two kernels (0), and the unfused
backward + Adam next to them. Per-stage HIP-event times (ms per call), S2 by default."""
import statistics
import sys

import torch

sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from FasterGSCudaBackend import FusedRasterizerOptimizer
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
from harness.scenes import make_garden_like, orbit_views

be = default_backend(); dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
g = T.Gaussians(make_garden_like(n), dev)
g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in orbit_views(8)]
S = [T.extract_settings(v, 16, v.background_color) for v in views]
tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in views]
res = {}
for rnd in range(3):
    for mode in (1, 0):
        be.lib.fgs_debug_set_option(3, mode)
        fo = FusedRasterizerOptimizer([getattr(g, k).detach().clone() for k in T.PARAM_ORDER], [8e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3])
        for i in range(2):
            fo.render_and_step(S[i], lambda img, i=i: be.l1_dssim(img, tg[i], 0.8, 0.2)[1], g.densification_info)
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
        for i in range(8):
            fo.render_and_step(S[i], lambda img, i=i: be.l1_dssim(img, tg[i], 0.8, 0.2)[1], g.densification_info)
        t1.record(); torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        res.setdefault(mode, []).append((t0.elapsed_time(t1) / 8, {k: round(t / 8, 4) for k, (t, c) in pr.items() if c > 0 and k in ('preprocess_backward', 'sh_rest_backward', 'fused_backward_adam', 'blend_backward')}))
        del fo
    # unfused
    for i in range(2): T.training_iteration(g, views[i], tg[i], i)
    torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    for i in range(8): T.training_iteration(g, views[i], tg[i], 2 + i)
    t1.record(); torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
    res.setdefault('unfused', []).append((t0.elapsed_time(t1) / 8, {k: round(t / 8, 4) for k, (t, c) in pr.items() if c > 0 and k in ('preprocess_backward', 'sh_rest_backward', 'adam', 'blend_backward')}))
be.lib.fgs_debug_set_option(3, 1)
for k, v in res.items():
    print('fused single kernel' if k == 1 else ('fused two kernels' if k == 0 else k), 'median step ms', round(statistics.median(x[0] for x in v), 4), v[-1][1])
