"""The inference path (rasterize -> fgs_inference: preprocess_kernel<true>, blend_kernel<false>) on the fuzz configurations of the suite (tests/helpers.fuzz_configuration:
ragged counts, odd image sizes, every SH degree, both antialiasing modes, Gaussians on the cuts) against the oracle's inference mode, all four output forms (CHW / HWC,
clamped or not). Pixels the oracle's threshold-risk masks name are excluded and counted, as in the training-path fuzz tests. usage: python tools/inference_sweep.py A B"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers
from oracle import oracle as O
from FasterGSCudaBackend import rasterize
O.build()
a, b = int(sys.argv[1]), int(sys.argv[2])
worst, masked_total, bad = 0.0, 0, []
for seed in range(a, b):
    p, view, K, aa, label = helpers.fuzz_configuration(seed)
    S, RS = helpers.settings_pair(view, K, aa, device='cuda')
    dp = {k: v.cuda() for k, v in p.items()}
    ft = O.forward(*helpers.np_params(p), S, bucket_size=64)                        # training-mode run of the oracle: the risk masks are defined on it
    pm = helpers.flip_masks(O, ft, S)['pixel']
    masked_total += int(pm.sum())
    for to_chw, clamp in ((True, True), (True, False), (False, True), (False, False)):
        img = rasterize(*[dp[k] for k in helpers.NAMES], RS, to_chw, clamp).cpu().numpy()
        f = O.forward(*helpers.np_params(p), S, inference=True, to_chw=to_chw, clamp_output=clamp)
        assert img.shape == f['image'].shape, (label, img.shape, f['image'].shape)
        err = np.abs(img.astype(np.float64) - f['image'])
        err = err.max(axis=0) if to_chw else err.max(axis=2)
        e = float(np.where(pm, 0.0, err).max() / max(1.0, float(np.abs(f['image']).max())))
        worst = max(worst, e)
        if e >= 1e-4:
            bad.append((label, to_chw, clamp, e))
    if (seed - a + 1) % 200 == 0:
        print(f'{seed - a + 1} configurations: worst error outside the masks {worst:.2e}, masked pixels so far {masked_total}, beyond 1e-4: {len(bad)}', flush=True)
print(f'{b - a} configurations x 4 output forms: worst error outside the masks {worst:.2e}; beyond 1e-4: {len(bad)}')
for x in bad[:20]:
    print('  ', x)
