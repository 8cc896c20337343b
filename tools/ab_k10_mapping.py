"""K10's tile -> workgroup mappings in ONE process (fgs_debug_set_option(10, m)): 252 = static strips (default), 254 = device-side block plan (250, the strips as per-XCD queues
with stealing, existed for one commit of round 5 and measured slower: profiles/r05_ab_k10_mapping.txt). Scenes: S2, the layered S2 (opacity logits - 3), bench.py's surface scene (2 M thin disks on
surfaces, object-centric: the margins of the image are empty). Prints the blend_forward stage (training forward and inference), ms per launch, best of 4 rounds."""
import math, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
from harness.scenes import look_at_view, make_surface_scene
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
layered = dict(params); layered['opacities'] = params['opacities'] - 3.0
sviews = [look_at_view((6.4 * math.cos(2 * math.pi * k / 8), -(1.0 + 1.6 * (k % 3)), 6.4 * math.sin(2 * math.pi * k / 8)), (0.0, 1.3, 0.0), 1920, 1080, 1420.0) for k in range(8)]
import os
scenes = [('S2', params, views), ('layered S2', layered, views), ('surface 2 M', make_surface_scene(2_000_000), sviews)]
if os.environ.get('FGS_SCENES'): scenes = [s for s in scenes if s[0] in os.environ['FGS_SCENES'].split(',')]
modes = [(252, 'static strips')] if not hasattr(be.lib, 'fgs_debug_set_option') else [(252, 'static strips'), (254, 'device-side block plan')]
for name, p, vs in scenes:
    g = T.Gaussians(p, dev)
    vv = [v.to(dev) for v in vs[:4]]
    S = [T.extract_settings(v, g.active_sh_bases, v.background_color) for v in vv]
    P = g.tensors()
    ref = None
    for m, label in modes:
        if hasattr(be.lib, 'fgs_debug_set_option'):
            assert be.lib.fgs_debug_set_option(10, m) == 0
        best_t, best_i = 1e9, 1e9
        for rnd in range(4):
            be.profile_enable(True); be.profile_read()
            for s in S: res = be.forward(*P, s)
            torch.cuda.synchronize(); t = be.profile_read()['blend_forward'][0] / len(S)
            for s in S: img = be.inference(*P, s, True, True)
            torch.cuda.synchronize(); i = be.profile_read()['blend_forward'][0] / len(S)
            be.profile_enable(False)
            best_t, best_i = min(best_t, t), min(best_i, i)
        same = True if ref is None else bool(torch.equal(res.image, ref))
        ref = res.image.clone() if ref is None else ref
        print(f'{name:12s} {m} {label:26s} training blend {best_t:.4f} ms   inference blend {best_i:.4f} ms   image identical to the first mapping: {same}', flush=True)
    del g
if hasattr(be.lib, 'fgs_debug_set_option'):
    be.lib.fgs_debug_set_option(10, 252)
