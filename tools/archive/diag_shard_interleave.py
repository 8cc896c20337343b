import sys; sys.path[:0]=['/root/repo','/root/repo/faster-gaussian-splatting_amd','/root/repo/tests']
import numpy as np, torch, helpers
from harness.scenes import make_garden_like, orbit_views
from harness.distributed import SEGMENTS, ViewParallelTrainer
from harness import sharded as SH
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); DEV='cuda'
LRS = {'means': 1.6e-4, 'sh_coefficients_0': 2.5e-3, 'sh_coefficients_rest': 1.25e-4, 'opacities': 2.5e-2, 'scales': 5e-3, 'rotations': 1e-3}
params = {k: v.to(DEV) for k, v in make_garden_like(40_000).items()}
params['scales'] = params['scales'] + 0.7
views = orbit_views(8, width=640, height=360, focal=473.0)[:4]
RS = [helpers.settings_pair(v, device=DEV)[1] for v in views]
targets = [torch.rand(3, 360, 640, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(4)]
def run(interleave, steps=1):
    SH.INTERLEAVE_SHARDS = interleave
    grp = SH.LocalShardGroup(be, params, LRS, 4, fused=False)
    imgs = None
    for _ in range(steps):
        imgs = grp.step(RS, targets)
    g = {k: torch.cat([t.grads[k] for t in grp.ranks]) for k in SEGMENTS}
    return imgs, g, grp.gather_parameters()
i0, g0, p0 = run(False); i1, g1, p1 = run(True); i2, g2, p2 = run(True)
for v in range(4): print('image diff view', v, float((i0[v]-i1[v]).abs().max()), 'pixels differing', int(((i0[v]-i1[v]).abs().amax(0) > 0).sum()))
for k in SEGMENTS:
    d = (g0[k]-g1[k]).abs().max() / g0[k].abs().max(); d2 = (g2[k]-g1[k]).abs().max() / g1[k].abs().max()
    print(k, 'grad rel diff interleave vs not', float(d), ' run-to-run', float(d2))
# depth ties
SH.INTERLEAVE_SHARDS = True
