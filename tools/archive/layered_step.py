"""A few training iterations on the layered scene (S2 with opacity logits lowered by `shift`), for rocprofv3 runs of the blend kernels."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from harness import trainer as T
from harness.scenes import make_garden_like, orbit_views
shift = float(sys.argv[1]) if len(sys.argv) > 1 else -3.0
dev = torch.device('cuda:0')
p = make_garden_like(3_000_000); p['opacities'] = p['opacities'] + shift
g = T.Gaussians(p, dev); g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in orbit_views(8)]
tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in views[:4]]
for i in range(4): T.training_iteration(g, views[i], tg[i], i)
torch.cuda.synchronize()
