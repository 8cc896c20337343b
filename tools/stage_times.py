"""Stage times of the training iteration on S2 and on the layered scene (S2, opacity logits - 3) for the library named by FGS_HIP_LIBRARY (default:
the current build) -- the process-level half of tools/ab_two_libs.sh. Prints one line per scene: ms / iteration and the stages named on the
command line (default: the blend kernels; `all` = every stage and their sum)."""
import sys, time, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
keys = sys.argv[1:] or ['blend_forward', 'stage_pixels', 'blend_backward']
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
for shift in (0.0, -3.0):
    p2 = dict(params); p2['opacities'] = params['opacities'] + shift
    g = T.Gaussians(p2, dev); g.training_setup(training_cameras_extent=5.0)
    vs = [v.to(dev) for v in views]
    tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in vs]
    for i in range(3): T.training_iteration(g, vs[i], tg[i], i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16): T.training_iteration(g, vs[i % 8], tg[i % 8], 3 + i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 16 * 1e3
    be.profile_enable(True); be.profile_read()
    for i in range(8): T.training_iteration(g, vs[i], tg[i], 20 + i)
    torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
    ks = [k for k in pr if pr[k][1] > 0] if keys == ['all'] else keys
    print(f'shift {shift:4.1f}  {ms:.3f} ms/it  ' + '  '.join(f'{k} {pr[k][0] / 8:.4f}' for k in ks) + (f'  sum {sum(pr[k][0] for k in ks) / 8:.3f}' if keys == ['all'] else ''), flush=True)
    del g, tg
