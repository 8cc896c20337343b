"""A/B of the depth-sort variants inside one process (fgs_debug_set_option(9, m)): 0 = round 1 (4 x 8 bits over all 32 bits, 4096-item
workgroups), 1 = key - bits(near) in 9-bit passes (27 bits = 3 passes for near 0.2 / far 1e4), 2 = 2048-item workgroups, 3 = both; 1 is the default.
Stage time of the depth sort and the whole frame, 8 views of S2, three rounds."""
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
print('near / far of the views:', S[0].near_plane, S[0].far_plane)
res = {}
for rnd in range(3):
    for mode in (0, 1, 2, 3):
        assert be.lib.fgs_debug_set_option(9, mode) == 0
        for s in S[:2]: be.inference(*g.tensors(), s, True, True)
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
        for s in S: be.inference(*g.tensors(), s, True, True)
        t1.record(); torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        res.setdefault(mode, []).append((round(pr['depth_sort'][0] / 8, 4), round(t0.elapsed_time(t1) / 8, 4)))
be.lib.fgs_debug_set_option(9, 1)
names = {0: 'round 1: 4 x 8 bits, 4096 items', 1: 'key range (9-bit passes), 4096 items (default)', 2: '4 x 8 bits, 2048 items', 3: 'key range + 2048 items'}
for mode, v in res.items(): print(f'{names[mode]:42s} (depth_sort ms, frame ms):', v)
