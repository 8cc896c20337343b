"""A/B of the blend-backward variants inside ONE process (interleaved rounds; median of HIP-event times per launch)."""
import sys, statistics, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from harness.scenes import make_garden_like, orbit_views
from harness import trainer as T
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
g = T.Gaussians(make_garden_like(n), dev)
variants = [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '2,3'.split(','))]
res = {v: [] for v in variants}
views = [v.to(dev) for v in orbit_views(8)]
for rnd in range(6):
    v = views[rnd % 8]
    S = T.extract_settings(v, 16, v.background_color)
    fw = be.forward(*g.tensors(), S)
    gi = torch.randn_like(fw.image) / fw.image.numel()
    for var in variants:
        be.lib.fgs_debug_set_backward_variant(var)
        args = (torch.empty(0, device=dev), gi, fw.image, g.means, g.scales, g.rotations, g.opacities, g.sh_coefficients_rest, fw.buffers, S, fw.state)
        be.backward(*args); torch.cuda.synchronize()
        be.profile_enable(True); be.profile_read()
        for _ in range(3): be.backward(*args)
        torch.cuda.synchronize(); t, c = be.profile_read()['blend_backward']; be.profile_enable(False)
        res[var].append(t / c)
be.lib.fgs_debug_set_backward_variant(3)
for var in variants: print('variant', var, 'median ms', round(statistics.median(res[var]), 4), 'min', round(min(res[var]), 4), 'max', round(max(res[var]), 4))
