"""One fuzz configuration (tests/helpers.fuzz_configuration(seed)) on the GPU against the oracle: where is the largest image error outside the
threshold-risk mask, and at which width of the alpha / transmittance band (helpers.flip_masks eps) would the oracle have named that pixel?
A flip just outside the band is arithmetic (device exp / log against glibc); a pixel no band names is a bug. usage: python tools/diag_fuzz_seed.py SEED"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers
from oracle import oracle as O
from FasterGSCudaBackend._backend import default_backend
O.build()
if sys.argv[1].startswith('mid:'):                       # mid:SEED = the mid-scale configurations of tests/test_gpu_fuzz.py (1e5 Gaussians)
    import test_gpu_fuzz
    seed = int(sys.argv[1][4:]); p, view, K, aa, label = test_gpu_fuzz._mid_scale_configuration(seed)
else:
    seed = int(sys.argv[1]); p, view, K, aa, label = helpers.fuzz_configuration(seed)
print(label)
be = default_backend(); dev = torch.device('cuda:0')
S, RS = helpers.settings_pair(view, K, aa, device=dev)
dp = {k: v.to(dev) for k, v in p.items()}
res = be.forward(*[dp[k] for k in helpers.NAMES], RS)
f = O.forward(*helpers.np_params(p), S, bucket_size=64)
n = dp['means'].shape[0]
dec = helpers.decode_forward(be, res, n, view.width, view.height)
img = res.image.cpu().numpy(); ref = f['image']
err = np.abs(img.astype(np.float64) - ref).max(axis=0)
m0 = helpers.flip_masks(O, f, S, dec)['pixel']
e2 = np.where(m0, 0.0, err)
y, x = np.unravel_index(np.argmax(e2), e2.shape)
print(f'largest unmasked error {e2.max():.3e} at pixel ({x}, {y}); device {img[:, y, x]} oracle {ref[:, y, x]}')
npr = helpers.tiles_to_image(dec['n_processed_tiles'], view.width, view.height)
print('last contributor: device', int(npr[y, x]), 'oracle', int(f['n_processed'].reshape(npr.shape)[y, x]))
print('pixels beyond 1e-4 outside the mask:', int((e2 > 1e-4).sum()))
for eps in (5e-6, 1e-5, 2e-5, 5e-5, 1e-4, 1e-3):
    m = helpers.flip_masks(O, f, S, dec, eps=eps, eps_T=max(1e-5, eps))['pixel']
    print(f'band {eps:.0e}: {int(m.sum())} pixels named, this pixel named: {bool(m[y, x])}, pixels beyond 1e-4 outside: {int((np.where(m, 0.0, err) > 1e-4).sum())}')
# the device's own records against the oracle's (opacity / conic of the visible Gaussians)
for k in ('conic_opacity', 'mean2d'):
    if k in dec and k in f:
        a, b = np.asarray(dec[k], np.float64), np.asarray(f[k], np.float64)
        vis = f['n_touched'] > 0
        d = np.abs(a[vis] - b[vis]) / np.maximum(np.abs(b[vis]), 1e-30)
        print(k, 'max relative difference over visible Gaussians', float(d.max()))

# the tile list of that pixel, entry by entry, from the oracle's records and from the device's
gw = (view.width + 15) // 16
tile = (y // 12) * gw + (x // 16)
sx0, sy0 = (x // 8) * 8, (y // 4) * 4                     # the pixel's 8 x 4 sub-tile
def walk(src, name):
    r0, r1 = [int(v) for v in src['ranges'][tile]]
    print(f'-- {name}: tile {tile} holds instances [{r0}, {r1})')
    T = 1.0
    for i in range(r0, r1):
        g = int(src['inst_prims'][i])
        m, co, sb = src['mean2d'][g].astype(np.float64), src['conic_opacity'][g].astype(np.float64), [int(v) for v in src['screen_bounds'][g]]
        dx, dy = m[0] - 0.5 - x, m[1] - 0.5 - y                      # pixel centre convention of the blend (kf:445-470): see oracle
        power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
        alpha = co[3] * np.exp(min(power, 0.0))
        in_sub = sb[0] < sx0 + 8 and sb[1] > sx0 and sb[2] < sy0 + 4 and sb[3] > sy0
        in_px = sb[0] <= x < sb[1] and sb[2] <= y < sb[3]
        if r1 - r0 <= 12 or abs(alpha * 255 - 1.0) < 2e-3 or (in_sub and alpha >= 1.0 / 255.0 and T < 3e-4):      # long lists: the borderline entries only
            print(f'   entry {i - r0}: Gaussian {g} bounds {sb} overlaps sub-tile {in_sub} holds pixel {in_px} alpha {alpha:.6e} (x 255 = {alpha * 255:.6f}) T before {T:.5f}')
        if in_sub and alpha >= 1.0 / 255.0:
            T *= 1.0 - alpha
            if T < 1e-4:
                print(f'   (transmittance test ends the walk behind entry {i - r0}: T = {T:.6e})'); break
walk(f, 'oracle records')
walk(dec, 'device records')

# backward: which opacity-gradient entries miss the element-wise bar against the fp64 values, on the device and in the fp32 oracle
gi = np.random.default_rng(3).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
g = O.backward(f, S, gi, np.zeros((2, n), np.float32))
grads = be.backward(torch.zeros(2, n, device=dev), torch.from_numpy(gi).to(dev), res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'],
                    dp['sh_coefficients_rest'], res.buffers, RS, res.state)
got = {k: t.cpu().numpy() for k, t in zip(helpers.GRAD_KEYS, grads)}
t64 = O.forward_backward_f64(f, S, gi)
masks = helpers.flip_masks(O, f, S, dec)
key = [k for k in helpers.GRAD_KEYS if 'opac' in k][0]
a, r32, tr = got[key].reshape(-1).astype(np.float64), g[key].reshape(-1).astype(np.float64), np.asarray(t64[key]).reshape(-1)
keep = ~masks['prim'] & ~masks['near']
med = np.median(np.abs(tr[keep]))
bar = 1e-4 * np.abs(tr) + 1e-4 * med
miss_h, miss_o = keep & (np.abs(a - tr) > bar), keep & (np.abs(r32 - tr) > bar)
print(f'opacity gradient: {int(keep.sum())} entries kept, median |value| {med:.3e}; beyond the bar: device {int(miss_h.sum())}, oracle32 {int(miss_o.sum())}')
op = 1.0 / (1.0 + np.exp(-p['opacities'].numpy().reshape(-1).astype(np.float64)))
for j in np.where(miss_h | miss_o)[0]:
    print(f'   Gaussian {j}: fp64 {tr[j]: .6e} device {a[j]: .6e} (err {abs(a[j]-tr[j]):.2e}) oracle32 {r32[j]: .6e} (err {abs(r32[j]-tr[j]):.2e}) bar {bar[j]:.2e} | tiles {int(f["n_touched"][j])} '
          f'opacity {op[j]:.5f} (x255 = {op[j]*255:.4f}) bounds {[int(v) for v in f["screen_bounds"][j]]}')

# how many visible Gaussians have a device record that differs from the oracle's by more than r (relative, worst of conic a b c and opacity)
both = (f['n_touched'] > 0) & (dec['n_touched'] > 0)
aa_, bb_ = np.asarray(dec['conic_opacity'], np.float64)[both], np.asarray(f['conic_opacity'], np.float64)[both]
relc = np.abs(aa_ - bb_) / np.maximum(np.abs(bb_), 1e-30)
print('records differing by more than r:', {r: (int((relc[:, :3].max(axis=1) > r).sum()), int((relc[:, 3] > r).sum())) for r in (1e-6, 1e-5, 3e-5, 1e-4, 1e-3)}, '(conic, opacity) of', int(both.sum()), 'visible')
