"""Phase shares of K11 variant 4 (library built by tools/k11m_phases.sh build; FGS_HIP_LIBRARY points at it): S2 and the layered scene, one view each."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
dev = torch.device('cuda:0'); be = default_backend()
raw = C.CDLL(os.environ['FGS_HIP_LIBRARY'])
raw.fgs_debug_k11m_phases.argtypes = [C.c_void_p, C.c_int]
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
be.lib.fgs_debug_set_backward_variant(4)
for shift in (0.0, -3.0):
    p2 = dict(params); p2['opacities'] = params['opacities'] + shift
    g = T.Gaussians(p2, dev)
    v = views[2].to(dev)
    S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    P = g.tensors()
    gi = torch.randn(3, v.height, v.width, device=dev) / (3 * v.height * v.width)
    res = be.forward(*P, S)
    run = lambda: (be.backward(None, gi, res.image, P[0], P[1], P[2], P[3], P[5], res.buffers, S, res.state), torch.cuda.synchronize())
    run(); run()
    assert raw.fgs_debug_k11m_phases(None, 1) == 0
    be.profile_enable(True); be.profile_read(); run(); pr = be.profile_read(); be.profile_enable(False)
    ph = np.zeros(8, np.uint64)
    assert raw.fgs_debug_k11m_phases(ph.ctypes.data, 0) == 0
    ph = ph.astype(np.float64)
    tot = ph[1:6].sum()
    print(f'opacity shift {shift}: kernel {pr["blend_backward"][0] / pr["blend_backward"][1]:.3f} ms (with probes); item-waves {ph[0]:.0f}, pairs {ph[6]:.0f} ({ph[6] / ph[0]:.1f} per item-wave), matrix passes {ph[7]:.0f}')
    print('  wave cycles per item-wave: ' + '  '.join(f'{n} {ph[i] / ph[0]:.0f} ({100 * ph[i] / tot:.0f} %)' for i, n in ((1, 'staging'), (2, 'cull'), (3, 'walk'), (4, 'matrix passes'), (5, 'tail'))))
    print(f'  walk: {ph[3] / max(ph[6], 1):.0f} cycles per pair; matrix pass: {ph[4] / max(ph[7], 1):.0f} cycles each', flush=True)
be.lib.fgs_debug_set_backward_variant(3)
