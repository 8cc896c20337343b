"""Are two forward passes of the same inputs bit-identical on hardware? (Equal depth keys keep the run-to-run varying order of K1's atomic
compaction through the stable sort, exactly as in the reference, kf:204-208.)"""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
for shift in (0.0, -3.0):
    p2 = dict(params); p2['opacities'] = params['opacities'] + shift
    g = T.Gaussians(p2, dev)
    v = views[3].to(dev); S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    imgs = [be.forward(*g.tensors(), S).image.clone() for _ in range(4)]
    d = [(imgs[0] - im).abs().max().item() for im in imgs[1:]]
    print(f'opacity shift {shift}: max |difference| between repeated forward passes (same library, static mapping): {d}; pixels that differ: {[int((imgs[0] != im).any(0).sum()) for im in imgs[1:]]}')
