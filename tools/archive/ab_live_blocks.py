"""A/B inside one process: the reference-order training iteration (render -> loss -> backward -> FusedAdam.step) with and without the
live-block hand-over (FasterGSCudaBackend.set_live_block_handover): ms per iteration, Adam stage time, and that the flags are matched."""
import sys, time, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
import FasterGSCudaBackend as FGS
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in views]
targets = [torch.rand(3, v.height, v.width, device=dev) for v in views]
res = {}
it = 0
for rnd in range(3):
    for mode in (False, True):
        FGS.set_live_block_handover(mode)
        for i in range(3): T.training_iteration(g, views[i % 8], targets[i % 8], it); it += 1
        torch.cuda.synchronize(); s0 = FGS.live_block_stats()
        t0 = time.perf_counter()
        for i in range(16): T.training_iteration(g, views[i % 8], targets[i % 8], it); it += 1
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16 * 1e3
        be.profile_enable(True); be.profile_read()
        for i in range(8): T.training_iteration(g, views[i % 8], targets[i % 8], it); it += 1
        torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        s1 = FGS.live_block_stats()
        res.setdefault(mode, []).append((round(dt, 4), round(pr['adam'][0] / 8, 4), round(pr['preprocess_backward'][0] / 8, 4), s1['matched'] - s0['matched'], s1['missed'] - s0['missed']))
for mode, v in res.items(): print('hand-over', 'ON ' if mode else 'off', '(ms/iteration, adam ms, K12 ms, matched, missed):', v)
