"""Heavy hitters of K11's atomics: per view, the largest per-Gaussian tile counts and how close their indices are."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd', '/root/repo/tests']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev)
ORDER = ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')
P = {k: getattr(g, k).detach() for k in ORDER}
n = P['means'].shape[0]
for vi, v in enumerate(views):
    v = v.to(dev); S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    res = be.forward(*(P[k] for k in ORDER), S)
    lay = be.blob_layout(0, n, v.width, v.height, 0, 0)
    nt = be.view(res.buffers[0], lay, 'n_touched', torch.int32)[:n].long()
    top = torch.topk(nt, 12)
    idx = top.indices.tolist()
    order = sorted(range(len(idx)), key=lambda a: idx[a])
    print(vi, 'top tiles', top.values.tolist(), 'idx', idx, '#>256:', int((nt > 256).sum()), '#>1024:', int((nt > 1024).sum()),
          'same-line pairs among >256:', int(((torch.sort(torch.nonzero(nt > 256).flatten() // 32).values.diff() == 0).sum())))
