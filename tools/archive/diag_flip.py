"""Diagnostics for the flip-aware parity tests: error statistics of the HIP path against the oracle with and without the threshold masks."""
import sys
import numpy as np
import torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd', '/root/repo/tests']
import helpers
from oracle import oracle as O
from FasterGSCudaBackend._backend import default_backend
from harness.scenes import make_garden_like, orbit_views
O.build()
be = default_backend()
DEV = 'cuda'
cases = [('60k', 60_000, dict(width=640, height=360, focal=473.0), 1, 0.7), ('S1', 1_000_000, {}, 0, 0.0), ('S2', 3_000_000, {}, 3, 0.0)]
for label, n, vkw, vi, ds in cases:
    params = make_garden_like(n)
    params['scales'] = params['scales'] + ds
    view = orbit_views(8, **vkw)[vi]
    S, RS = helpers.settings_pair(view, device=DEV)
    dp = {k: v.to(DEV).contiguous() for k, v in params.items()}
    res = be.forward(*[dp[k] for k in helpers.NAMES], RS)
    f = O.forward(*helpers.np_params(params), S, bucket_size=64)
    dec = helpers.decode_forward(be, res, n, view.width, view.height)
    print(label, 'V', dec['V'], f['V'], 'I', dec['I'], f['I'], 'n_touched mismatches', int((dec['n_touched'] != f['n_touched']).sum()),
          'bounds mismatches', int((dec['screen_bounds'] != f['screen_bounds']).any(axis=1)[f['n_touched'] > 0].sum()))
    gi = np.random.default_rng(3).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    dens_o = np.zeros((2, n), np.float32)
    g = O.backward(f, S, gi, dens_o)
    dens = torch.zeros(2, n, device=DEV)
    grads = be.backward(dens, torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'],
                        dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    img = res.image.cpu().numpy()
    err_img = np.abs(img.astype(np.float64) - f['image']).max(axis=0)
    print('  image: max err', err_img.max(), 'pixels > 1e-4:', int((err_img > 1e-4).sum()), '> 1e-5:', int((err_img > 1e-5).sum()))
    for eps, eps_T in ((1e-5, 1e-4), (1e-5, 0.0), (3e-5, 0.0), (1e-4, 0.0)):
        r = O.threshold_risk(f, S, eps, eps_T)
        pm, gm, near = r['pixel'], r['prim'], r['near']
        bad_px = err_img > 1e-4
        print(f'  eps {eps:g} eps_T {eps_T:g}: masked px {pm.mean():.2e} prims {gm.mean():.2e} near {near.mean():.2e}; bad pixels unmasked {int((bad_px & ~pm).sum())} of {int(bad_px.sum())}; unmasked image err {err_img[~pm].max():.2e}')
        for k, t in zip(helpers.GRAD_KEYS, grads):
            a = t.cpu().numpy().reshape(g[k].shape).astype(np.float64)
            e = np.abs(a - g[k]).reshape(n, -1).max(axis=1) / (np.abs(g[k]).max() + 1e-30)
            bad = e > 1e-4
            print(f'     {k:10s} rel_inf all {e.max():.2e} unmasked {e[~gm].max():.2e} excluding near too {e[~gm & ~near].max():.2e}; bad {int(bad.sum())} unmasked bad {int((bad & ~gm).sum())} of which near {int((bad & ~gm & near).sum())}')
    del res, grads
