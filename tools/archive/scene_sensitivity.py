"""Stage times of one training iteration as the scene gets less opaque (more Gaussians blended per pixel): how the kernel
ranking of S2 shifts on scenes with deeper semi-transparent layering."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
import itertools
for off, variant in itertools.product((0.0, -1.5, -3.0), (3, 2)):
    be.lib.fgs_debug_set_backward_variant(variant)
    p = {k: v.clone() for k, v in params.items()}; p['opacities'] = p['opacities'] + off
    g = T.Gaussians(p, dev); g.training_setup(training_cameras_extent=5.0)
    V = [v.to(dev) for v in views]
    tg = [T.render_image_benchmark(g, v).clone() * 0.9 for v in V]
    for i in range(3): T.training_iteration(g, V[i], tg[i], i)
    torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    for i in range(8): T.training_iteration(g, V[i], tg[i], 3 + i)
    t1.record(); torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
    res = be.forward(*g.tensors(), T.extract_settings(V[0], 16, V[0].background_color))
    lay = be.blob_layout(1, g.means.shape[0], 1920, 1080, res.state[1], res.state[2])
    mx = be.view(res.buffers[1], lay, 'max_n_processed', torch.int32)[:10800].long()
    print(f'K11 variant {variant} opacity logit offset {off}: step {t0.elapsed_time(t1) / 8:.3f} ms, processed buckets/tile {float(((mx + 63) // 64).float().mean()):.2f}',
          {k: round(t / 8, 3) for k, (t, c) in pr.items() if c > 0 and k in ('preprocess', 'create_instances', 'tile_sort', 'blend_forward', 'blend_backward', 'adam')})
