"""A/B of K1's sequential-tile threshold inside one process (stage times from HIP events, all 8 views of S2)."""
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
    for L in (16, 0, 8):
        be.lib.fgs_debug_set_option(5, L)
        for s in S[:2]: be.inference(*g.tensors(), s, True, True)
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        for s in S: be.inference(*g.tensors(), s, True, True)
        torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        res.setdefault(L, []).append(pr['preprocess'][0] / pr['preprocess'][1])
for L, v in res.items(): print('seq_tiles', L, 'preprocess ms', [round(x, 4) for x in v])
be.lib.fgs_debug_set_option(5, 0)
