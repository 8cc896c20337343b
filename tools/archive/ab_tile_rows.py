"""A/B of the forward blend's tile -> workgroup mapping inside one process (fgs_debug_set_option(10, g)): 0 = one contiguous band of tile rows
per XCD (round 1), 255 = the same bands walked bottom-up, g >= 1 = groups of g rows dealt to the XCDs in turn, bottom of the image first. Training-forward and inference stage
times, S2 and the layered scene."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
for shift in (0.0, -3.0):
    p2 = dict(params); p2['opacities'] = params['opacities'] + shift
    g = T.Gaussians(p2, dev)
    S = [T.extract_settings(v.to(dev), g.active_sh_bases, v.to(dev).background_color) for v in views]
    res = {}
    for rnd in range(4):
        for G in (0, 255, 1, 2):
            assert be.lib.fgs_debug_set_option(10, G) == 0
            for s in S[:2]: be.forward(*g.tensors(), s); be.inference(*g.tensors(), s, True, True)
            torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
            for s in S: be.forward(*g.tensors(), s)
            torch.cuda.synchronize(); tr = be.profile_read()['blend_forward'][0] / 8; be.profile_read()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            for s in S: be.inference(*g.tensors(), s, True, True)
            torch.cuda.synchronize(); inf = be.profile_read()['blend_forward'][0] / 8; be.profile_enable(False)
            res.setdefault(G, []).append((round(tr, 4), round(inf, 4)))
    print(f'opacity shift {shift}: (training blend ms, inference blend ms) per row-group size')
    for G, v in res.items(): print(f'   g = {G}: {v}')
be.lib.fgs_debug_set_option(10, 0)
