"""K11 on the records path (sharded multi-GPU renderer) vs the whole pipeline, per view, in one process (HIP-event stage times).
This comparison exposed the heavy-hitter line contention described in DESIGN.md section 6."""
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
rec = torch.empty((1, n, 56), dtype=torch.uint8, device=dev); cnt = torch.zeros((1, 2), dtype=torch.int32, device=dev)
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for vi, v in enumerate(views):
    v = v.to(dev); S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    tgt = be.inference(*(P[k] for k in ORDER), S, True, True) * 0.9
    def whole():
        res = be.forward(*(P[k] for k in ORDER), S)
        gl = be.l1_dssim(res.image, tgt, 0.8, 0.2, with_grad=True)[1]
        be.backward(None, gl, res.image, P['means'], P['scales'], P['rotations'], P['opacities'], P['sh_coefficients_rest'], res.buffers, S, res.state)
        return res.state
    def cut():
        be.shard_preprocess(*(P[k] for k in ORDER), [S], rec, cnt)
        V, I = cnt[0].tolist()
        res = be.forward_from_records(rec[0, :V].reshape(-1), V, I, S, 15)
        gl = be.l1_dssim(res.image, tgt, 0.8, 0.2, with_grad=True)[1]
        be.backward_to_records(gl, res.image, res.buffers, S, res.state, 15)
        return res.state
    out = {}
    for name, fn in (('whole', whole), ('cut', cut)):
        for _ in range(2): st = fn()
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        for _ in range(4): fn()
        torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        out[name] = {k: round(t / c, 3) for k, (t, c) in pr.items() if c > 0 and k in ('blend_backward',)}
    print(vi, 'V', st[0], 'I', st[1], out)
