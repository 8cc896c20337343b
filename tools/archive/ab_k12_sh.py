"""How much of K1 / K12 is the strided sh_rest access? Stage times with active SH degree 3 vs 0 (same scene, one process)."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev)
P = g.tensors()
for K in (16, 4, 1, 16):
    S = [T.extract_settings(v.to(dev), K, v.to(dev).background_color) for v in views]
    gi = torch.randn(3, 1080, 1920, device=dev) / (3 * 1080 * 1920)
    def run(s):
        res = be.forward(*P, s)
        be.backward(None, gi, res.image, P[0], P[1], P[2], P[3], P[5], res.buffers, s, res.state)
    for s in S[:2]: run(s)
    torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
    for s in S: run(s)
    torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
    print('active bases', K, {k: round(t / c, 4) for k, (t, c) in pr.items() if c > 0 and k in ('preprocess', 'preprocess_backward', 'sh_rest_backward')})
