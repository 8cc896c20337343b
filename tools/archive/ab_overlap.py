"""A/B of the two fork-join overlaps inside the library (fgs_debug_set_option(14, 0|1)): SH colour of the visible Gaussians on a second stream
during K2-K9, accumulator clearing during the pixel staging pass. One process, interleaved rounds, S2 and the layered scene: ms per training
iteration (wall clock, 16 iterations) and the stage times that change."""
import sys, time, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
for shift in (0.0, -3.0):
    p2 = dict(params); p2['opacities'] = params['opacities'] + shift
    g = T.Gaussians(p2, dev); g.training_setup(training_cameras_extent=5.0)
    vs = [v.to(dev) for v in views]
    tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in vs]
    res = {0: [], 1: []}
    for rnd in range(4):
        for ov in (1, 0):
            assert be.lib.fgs_debug_set_option(14, ov) == 0
            for i in range(3): T.training_iteration(g, vs[i], tg[i], i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(16): T.training_iteration(g, vs[i % 8], tg[i % 8], 3 + i)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 16 * 1e3
            t0 = time.perf_counter()
            for i in range(16): T.render_image_benchmark(g, vs[i % 8])
            torch.cuda.synchronize(); ms_inf = (time.perf_counter() - t0) / 16 * 1e3
            be.profile_enable(True); be.profile_read()
            for i in range(8): T.training_iteration(g, vs[i], tg[i], 20 + i)
            torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
            res[ov].append((round(ms, 3), round(ms_inf, 3), round(pr['preprocess'][0] / 8, 4), round(pr.get('sh_colour_overlapped', (0, 0))[0] / 8, 4),
                            round(pr['stage_pixels'][0] / 8, 4)))
    print(f'opacity shift {shift}: (ms / training iteration, ms / inference frame, preprocess, sh_colour on the side stream, stage_pixels)')
    for ov in (1, 0): print(f'   overlap {ov}: {res[ov]}')
be.lib.fgs_debug_set_option(14, 1)
