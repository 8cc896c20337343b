import sys
sys.path[:0] = ['/root/repo/tests', '/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import numpy as np, torch, helpers
import oracle.oracle as oracle
from FasterGSCudaBackend._backend import default_backend
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 126
p, view, K, aa, label = helpers.fuzz_configuration(seed)
S, RS = helpers.settings_pair(view, K, aa, device='cuda')
be = default_backend()
dp = {k: v.cuda() for k, v in p.items()}
res = be.forward(*[dp[k] for k in helpers.NAMES], RS)
f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
n = p['means'].shape[0]
dec = helpers.decode_forward(be, res, n, view.width, view.height)
img = res.image.cpu().numpy()
err = np.abs(img - f['image']).max(axis=0)
print(label, 'image max err', err.max(), 'rel', err.max() / max(1.0, np.abs(f['image']).max()))
npr = helpers.tiles_to_image(dec['n_processed_tiles'], view.width, view.height); fT = helpers.tiles_to_image(dec['final_T_tiles'], view.width, view.height)
onp = f['n_processed'].reshape(view.height, view.width); oT = f['final_T'].reshape(view.height, view.width)
ys, xs = np.unravel_index(np.argsort(err.ravel())[-5:], err.shape)
for y, x in zip(ys, xs):
    print('pixel', x, y, 'err', err[y, x], 'n_processed gpu/oracle', npr[y, x], onp[y, x], 'final_T', fT[y, x], oT[y, x])
print('n_touched mismatches', int((dec['n_touched'] != f['n_touched']).sum()), 'V', dec['V'], f['V'], 'I', dec['I'], f['I'])
for k in ('mean2d', 'conic_opacity', 'color'):
    vis = f['n_touched'] > 0
    print(k, 'rel_inf', helpers.rel_inf(dec[k][vis], f[k][vis]))
m = helpers.flip_masks(oracle, f, S, dec)
print('mask pixels', int(m['pixel'].sum()), 'prims', int(m['prim'].sum()))
# the Gaussians of the worst pixel
y, x = ys[-1], xs[-1]
tile = (y // 12) * ((view.width + 15) // 16) + x // 16
rr = np.asarray(f['ranges']).reshape(-1, 2); r0, r1 = int(rr[tile, 0]), int(rr[tile, 1])
co, m2 = f['conic_opacity'], f['mean2d']
T = 1.0
for j in range(int(onp[y, x]) + 2):
    if r0 + j >= r1: break
    pr = f['inst_prims'][r0 + j]
    dx, dy = np.float32(m2[pr, 0] - (x + 0.5)), np.float32(m2[pr, 1] - (y + 0.5))
    t1, t2, t3 = co[pr, 0] * dx * dx, co[pr, 2] * dy * dy, co[pr, 1] * dx * dy
    expo = -0.5 * (t1 + t2) - t3
    alpha = co[pr, 3] * np.exp(min(expo, 0.0))
    col = f['color'][pr]
    print(f'  j={j} prim={pr} color={col} alpha={alpha:.9f} alpha*255={alpha * 255:.7f} terms {t1:.3f} {t2:.3f} {t3:.3f} T={T:.6g} opacity={co[pr, 3]:.7f}')
    if alpha >= 1 / 255: T *= 1 - alpha
print('channels gpu   ', img[:, y, x]); print('channels oracle', f['image'][:, y, x], 'bg', view.background_color.numpy())
# fp64 replay of the oracle's own list for this pixel
C = np.zeros(3); T = 1.0; contributions = []
for j in range(int(onp[y, x])):
    pr = f['inst_prims'][r0 + j]
    sb = f['screen_bounds'][pr]
    sx0 = (x // 8) * 8; sy0 = (y // 4) * 4
    if not (sb[0] < sx0 + 8 and sx0 < sb[1] and sb[2] < sy0 + 4 and sy0 < sb[3]): continue
    dx, dy = np.float32(m2[pr, 0]) - np.float32(x + 0.5), np.float32(m2[pr, 1]) - np.float32(y + 0.5)
    expo = -0.5 * (float(co[pr, 0]) * dx * dx + float(co[pr, 2]) * dy * dy) - float(co[pr, 1]) * dx * dy
    alpha = float(co[pr, 3]) * np.exp(min(expo, 0.0))
    if alpha < 1 / 255: continue
    col = np.maximum(f['color'][pr].astype(np.float64), 0.0)
    C += T * alpha * col; contributions.append((j, pr, T * alpha, col)); T *= 1 - alpha
print('fp64 replay', C + T * view.background_color.numpy().astype(np.float64), 'T', T)
big = sorted(contributions, key=lambda c: -np.abs(c[3]).max())[:3]
print('largest colours among the contributors:', [(j, pr, float(w), c.tolist()) for j, pr, w, c in big])
