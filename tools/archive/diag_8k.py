"""Where does the 8K image differ from the oracle outside the threshold-risk mask? (round 5, tests/test_gpu_parity.py::test_large_images_against_oracle)"""
import sys; sys.path[:0]=['/root/repo','/root/repo/faster-gaussian-splatting_amd','/root/repo/tests']
import numpy as np, torch, helpers
from harness.scenes import make_garden_like, orbit_views
from oracle import oracle as O
from FasterGSCudaBackend._backend import default_backend
be = default_backend()
n = 300_000
p = make_garden_like(n)
v = orbit_views(8, width=7680, height=4320, focal=5680.0)[4]
S, RS = helpers.settings_pair(v, device='cuda')
dp = {k: t.cuda().contiguous() for k, t in p.items()}
res = be.forward(*[dp[k] for k in helpers.NAMES], RS); torch.cuda.synchronize()
f = O.forward(*helpers.np_params(p), S, bucket_size=64)
dec = helpers.decode_forward(be, res, n, v.width, v.height)
print('V', dec['V'], f['V'], 'I', dec['I'], f['I'])
vis = (f['n_touched'] > 0) | (dec['n_touched'] > 0)
dnt = vis & (dec['n_touched'] != f['n_touched'])
dsb = vis & (dec['screen_bounds'] != f['screen_bounds']).any(axis=1)
print('prims with different tile count', int(dnt.sum()), 'different bounds', int(dsb.sum()))
masks = helpers.flip_masks(O, f, S, dec)
img = res.image.cpu().numpy()
err = np.abs(img.astype(np.float64) - f['image']).max(axis=0)
pm = masks['pixel']
bad = np.argwhere((err > 1e-4) & ~pm)
print('pixels beyond 1e-4 outside the mask:', len(bad), 'max err', float(err[~pm].max()))
sb = f['screen_bounds'].astype(np.int64); sbd = dec['screen_bounds'].astype(np.int64)
diff = np.nonzero(dnt | dsb)[0]
cover = 0
for (y, x) in bad[:2000]:
    inside = ((np.minimum(sb[diff, 0], sbd[diff, 0]) // 16 * 16 <= x) & (x < (np.maximum(sb[diff, 1], sbd[diff, 1]) + 15) // 16 * 16) &
              (np.minimum(sb[diff, 2], sbd[diff, 2]) // 12 * 12 <= y) & (y < (np.maximum(sb[diff, 3], sbd[diff, 3]) + 11) // 12 * 12))
    cover += bool(inside.any())
print('of the first', min(len(bad), 2000), 'bad pixels,', cover, 'lie in the tile box of a Gaussian whose integer intermediates differ')
if len(bad):
    y, x = bad[np.argmax(err[tuple(bad.T)])]
    print('worst pixel', (int(x), int(y)), 'tile', int(x) // 16, int(y) // 12, 'err', float(err[y, x]))
    t = (int(y) // 12) * 480 + int(x) // 16
    r_o = f['ranges'][t]; r_d = dec['ranges'][t]
    lo = set(f['inst_prims'][r_o[0]:r_o[1]].tolist()); ld = set(dec['inst_prims'][r_d[0]:r_d[1]].tolist())
    print('tile list oracle', len(lo), 'hip', len(ld), 'only oracle', sorted(lo - ld)[:5], 'only hip', sorted(ld - lo)[:5])
    for g in sorted((lo ^ ld))[:3]:
        print('  prim', g, 'n_touched o/h', int(f['n_touched'][g]), int(dec['n_touched'][g]), 'bounds o', sb[g].tolist(), 'h', sbd[g].tolist(), 'opacity', float(f['conic_opacity'][g, 3]))
