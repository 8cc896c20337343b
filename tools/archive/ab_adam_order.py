"""A/B inside one process: Adam's workgroup order (forward / reversed) in the unfused training iteration -- does the memory-side cache still
hold the tail of the gradients the backward pass has just written?"""
import statistics, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
from harness.scenes import make_garden_like, orbit_views
be = default_backend(); dev = torch.device('cuda:0')
g = T.Gaussians(make_garden_like(3_000_000), dev); g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in orbit_views(8)]
tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in views]
res = {0: [], 1: []}
for rnd in range(4):
    for mode in (0, 1):
        be.lib.fgs_debug_set_option(8, mode)
        for i in range(2): T.training_iteration(g, views[i], tg[i], i)
        torch.cuda.synchronize(); be.profile_enable(True, only='adam'); be.profile_read()
        for i in range(8): T.training_iteration(g, views[i], tg[i], 2 + i)
        torch.cuda.synchronize(); t, c = be.profile_read()['adam']; be.profile_enable(False)
        res[mode].append(t / c)
be.lib.fgs_debug_set_option(8, 1)
for mode, v in res.items(): print('adam order', 'reversed' if mode else 'forward ', 'ms per launch', [round(x, 4) for x in v], 'median', round(statistics.median(v), 4))
