"""Fused iteration (BASELINE.json configs[3]) on S2, view 0, for the library named by FGS_HIP_LIBRARY: ms / iteration and the fused kernel's own
time from HIP events (tools/ab_fused.sh alternates builds)."""
import sys, time, torch
import os; ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, ROOT + '/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from FasterGSCudaBackend import FusedRasterizerOptimizer
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
v = views[0].to(dev)
tgt = T.render_image_benchmark(g, v).clone() * 0.9
fo = FusedRasterizerOptimizer([getattr(g, k).detach() for k in T.PARAM_ORDER], [1.6e-4 * 5.0, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3])
S = T.extract_settings(v, g.active_sh_bases, v.background_color)
grad_fn = lambda img: be.l1_dssim(img, tgt, 0.8, 0.2)[1]
for _ in range(3): fo.render_and_step(S, grad_fn, g.densification_info)
be.profile_enable(True, only='fused_backward_adam'); be.profile_read()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): fo.render_and_step(S, grad_fn, g.densification_info)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
pr = be.profile_read()
print(f'{ms:.3f} ms/it  fused kernel {pr["fused_backward_adam"][0] / max(pr["fused_backward_adam"][1], 1):.4f} ms', flush=True)
