"""Useful share of the work K10 / K11 issue (library built by tools/pair_stats.sh build; FGS_HIP_LIBRARY points at it). Scenes: S2, the layered
scene (S2, opacity logits - 3) and, with FGS_PLY, a trained export. Two views, one forward + backward pass each."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
dev = torch.device('cuda:0'); be = default_backend()
raw = C.CDLL(os.environ['FGS_HIP_LIBRARY'])
for f in (raw.fgs_debug_k10_pair_stats, raw.fgs_debug_k11_pair_stats):
    f.argtypes = [C.c_void_p, C.c_int]


def measure(tag, params, views, view_ids):
    g = T.Gaussians(params, dev)
    for vi in view_ids:
        v = views[vi].to(dev)
        S = T.extract_settings(v, g.active_sh_bases, v.background_color)
        P = g.tensors()
        gi = torch.randn(3, v.height, v.width, device=dev) / (3 * v.height * v.width)
        assert raw.fgs_debug_k10_pair_stats(None, 1) == 0 and raw.fgs_debug_k11_pair_stats(None, 1) == 0
        res = be.forward(*P, S)
        be.backward(None, gi, res.image, P[0], P[1], P[2], P[3], P[5], res.buffers, S, res.state)
        torch.cuda.synchronize()
        k10, k11 = np.zeros(8, np.uint64), np.zeros(8, np.uint64)
        assert raw.fgs_debug_k10_pair_stats(k10.ctypes.data, 0) == 0 and raw.fgs_debug_k11_pair_stats(k11.ctypes.data, 0) == 0
        k10, k11 = k10.astype(np.float64), k11.astype(np.float64)
        n_vis, n_inst = res.state[0], res.state[1]
        tiles, staged, pairs, mine, passed, offered = k10[:6]
        items, steps, body, elig, p11, trim = k11[:6]
        print(f'{tag} view {vi}: N {g.means.shape[0]}  visible {n_vis}  instances {n_inst}')
        print(f'  K10: instances staged {staged:.0f} ({staged / max(n_inst, 1):.3f} of all)  (Gaussian, strip) slots offered {offered:.0f}  pairs walked {pairs:.0f} '
              f'= {pairs / max(offered, 1):.3f} of offered = {pairs / max(staged, 1):.2f} strips per staged instance')
        print(f'       lanes of walked pairs: sub-tile hit and pixel alive {mine / max(pairs * 64, 1):.3f}, blended (alpha test passed) {passed / max(pairs * 64, 1):.3f}; '
              f'blended (pixel, Gaussian) pairs {passed:.0f} = {passed / max(staged * 192, 1):.4f} of staged instances x 192 pixels')
        print(f'  K11: work items {items:.0f}  steps {steps:.0f} ({steps / max(items, 1):.1f} per item, 63 of them fill)  steps whose contribution block ran {body / max(steps, 1):.3f}')
        print(f'       lane-steps: issued {steps * 64:.0f}, real pixel in front of its last contributor {elig / max(steps * 64, 1):.3f}, passed the alpha test {p11 / max(steps * 64, 1):.4f} '
              f'({p11:.0f}; K10 blended {passed:.0f})')
        print(f'       live pixels outside the union of their bucket\'s 64 screen bounds (the ceiling of trimming the pixel side by the bucket\'s bounds): {trim:.0f} = {trim / max(steps, 1):.4f} of the steps')
        print(f'       a lane = pixel walk of the same lists visits {pairs:.0f} (Gaussian, strip) pairs x 64 lanes = {pairs / max(steps, 1):.3f} of the lane-steps K11 issues', flush=True)
        del res
    del g


if not os.environ.get('FGS_PAIR_STATS_ONLY_PLY'):
    sys.argv = ['bench.py']
    params, views, _ = bench.build_scene(bench.parse())
    measure('S2', params, views, [0, 2])
    p2 = dict(params); p2['opacities'] = params['opacities'] - 3.0
    measure('layered (S2, logits - 3)', p2, views, [0, 2])
    del params, p2
if os.environ.get('FGS_PLY'):
    sys.argv = ['bench.py', '--ply', os.environ['FGS_PLY']]
    params, views, what = bench.build_scene(bench.parse())
    measure(what.split(':')[0], params, views, [0, 2])
