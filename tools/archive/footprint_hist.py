"""Distribution of candidate-tile counts (bounding-box tiles) of the visible Gaussians of S2: guides K1 / K5 thresholds."""
import sys, torch, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev); n = g.means.shape[0]
for vi in (0, 1):
    v = views[vi].to(dev); S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    res = be.forward(*g.tensors(), S)
    lay = be.blob_layout(0, n, v.width, v.height, 0, 0)
    rec = be.view(res.buffers[0], lay, 'rec', torch.int32).view(n, 12)
    nt = be.view(res.buffers[0], lay, 'n_touched', torch.int32)[:n]
    vis = nt > 0
    bx, by = rec[vis, 9].long() & 0xffffffff, rec[vis, 10].long() & 0xffffffff
    x0, x1, y0, y1 = bx & 0xffff, bx >> 16, by & 0xffff, by >> 16
    cand = ((x1 + 15) // 16 - x0 // 16) * ((y1 + 11) // 12 - y0 // 12)
    c = cand.cpu().numpy(); t = nt[vis].cpu().numpy()
    print('view', vi, 'visible', len(c), 'mean cand', c.mean(), 'mean touched', t.mean())
    for thr in (4, 8, 16, 24, 32, 48, 64, 80, 128, 256, 1024):
        print(f'  cand > {thr}: {100.0 * (c > thr).mean():.2f} %   share of candidate work above: {100.0 * c[c > thr].sum() / c.sum():.1f} %')
