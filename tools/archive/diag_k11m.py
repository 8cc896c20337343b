"""Which of K11's nine accumulators differ between variant 3 and variant 4 on a trained export (FGS_PLY), and for what kind of Gaussian?
Goes through the sharded entry points (shard_preprocess -> forward_from_records -> backward_to_records), whose output is K11's raw result."""
import os, sys
import numpy as np, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
dev = torch.device('cuda:0'); be = default_backend()
sys.argv = ['bench.py', '--ply', os.environ['FGS_PLY']]
params, views, what = bench.build_scene(bench.parse())
g = T.Gaussians(params, dev)
names = ('d mean.x', 'd mean.y', 'd conic.a', 'd conic.b', 'd conic.c', 'd opacity', 'd colour.r', 'd colour.g', 'd colour.b')
for vi in (0, 3):
    v = views[vi].to(dev)
    S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    n = g.means.shape[0]
    rec = torch.zeros((1, n, 56), dtype=torch.uint8, device=dev); cnt = torch.zeros((1, 2), dtype=torch.int32, device=dev)
    be.shard_preprocess(*[getattr(g, k).detach() for k in ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')], [S], rec, cnt)
    V, I = int(cnt[0, 0]), int(cnt[0, 1])
    records = rec[0, :V].contiguous()
    res = be.forward_from_records(records.view(-1), V, I, S, 15)
    gi = torch.randn(3, v.height, v.width, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / (3 * v.height * v.width)
    acc = {}
    for var in (3, 4):
        be.lib.fgs_debug_set_backward_variant(var)
        acc[var] = be.backward_to_records(gi, res.image, res.buffers, S, res.state, 15).clone()
    be.lib.fgs_debug_set_backward_variant(3)
    a3, a4 = acc[3].double(), acc[4].double()
    print(f'{what} view {vi}: V {V} I {I}')
    for c in range(9):
        d = (a3[:, c] - a4[:, c]).abs()
        print(f'  {names[c]:12s} max |v3| {float(a3[:, c].abs().max()):.3e}  max |v3 - v4| {float(d.max()):.3e}  entries with |diff| > 1e-4 |v3| + 1e-6 max: '
              f'{int((d > 1e-4 * a3[:, c].abs() + 1e-6 * a3[:, c].abs().max()).sum())}')
    r32 = records.view(torch.float32).view(V, 14).cpu().numpy(); ru = r32.view(np.uint32)
    score = ((a3 - a4).abs() / a3.abs().max(dim=0).values.clamp_min(1e-30)).max(dim=1).values
    for i in torch.topk(score, 6).indices.tolist():
        bx, by = ru[i, 9], ru[i, 10]
        print(f'  record {i}: mean2d ({r32[i, 0]:.2f}, {r32[i, 1]:.2f}) conic ({r32[i, 2]:.3e}, {r32[i, 3]:.3e}, {r32[i, 4]:.3e}) opacity {r32[i, 5]:.3f} '
              f'bounds x [{bx & 0xffff}, {bx >> 16}) y [{by & 0xffff}, {by >> 16}) tiles {ru[i, 13]}')
        print('    v3 ' + ' '.join(f'{float(x):+.4e}' for x in a3[i]))
        print('    v4 ' + ' '.join(f'{float(x):+.4e}' for x in a4[i]), flush=True)
