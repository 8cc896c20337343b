import sys, torch, json
sys.path[:0]=['/root/repo','/root/repo/faster-gaussian-splatting_amd']
from harness.scenes import make_garden_like, orbit_views
from harness import trainer as T
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev = torch.device('cuda:0')
g = T.Gaussians(make_garden_like(3_000_000), dev)
for vi, v in enumerate(orbit_views(8)):
    v = v.to(dev); S = T.extract_settings(v, 16, v.background_color)
    for _ in range(2): be.forward(*g.tensors(), S)
    torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
    for _ in range(5): res = be.forward(*g.tensors(), S)
    torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
    print(vi, res.state, {k: round(t/c,3) for k,(t,c) in pr.items() if c})
