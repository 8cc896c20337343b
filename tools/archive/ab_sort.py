"""A/B of the binning sorts inside one process: rocPRIM onesweep (option 6 = 0) vs csrc/radix_sort.hip (1); stage times, 8 views of S2."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev)
S = [T.extract_settings(v.to(dev), g.active_sh_bases, v.to(dev).background_color) for v in views]
res = {}
for rnd in range(3):
    for impl in (0, 1, 3):
        be.lib.fgs_debug_set_option(6, impl)
        for s in S[:2]: be.inference(*g.tensors(), s, True, True)
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
        for s in S: be.inference(*g.tensors(), s, True, True)
        t1.record(); torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        res.setdefault(impl, []).append((round(pr['depth_sort'][0] / 8, 4), round(pr['tile_sort'][0] / 8, 4), round(t0.elapsed_time(t1) / 8, 4)))
for impl, v in res.items(): print({0: 'rocprim both', 1: 'own tile sort', 3: 'own tile + depth sort'}[impl], '(depth_sort, tile_sort, frame ms):', v)
