"""update_pruning_scores over the suite's fuzz configurations on the GPU against orc_pruning_scores (kernels_pruning_scores.cuh:348-505): scores outside the oracle's
threshold-risk masks (Gaussians with a borderline pair of their own, and -- looser -- Gaussians behind one) to 1e-4 of the largest score. usage: python tools/pruning_sweep.py A B"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers
from oracle import oracle as O
from FasterGSCudaBackend import update_pruning_scores
O.build()
a, b = int(sys.argv[1]), int(sys.argv[2])
worst, worst_near, bad = 0.0, 0.0, []
for seed in range(a, b):
    p, view, K, aa, label = helpers.fuzz_configuration(seed)
    n = p['means'].shape[0]
    S, RS = helpers.settings_pair(view, K, aa, device='cuda')
    ref = np.zeros(n, np.float32)
    O.pruning_scores(ref, *helpers.np_params(p), S)
    scores = torch.zeros(n, device='cuda')
    update_pruning_scores(scores, *[p[k].cuda() for k in helpers.NAMES], RS)
    got = scores.cpu().numpy()
    m = helpers.flip_masks(O, O.forward(*helpers.np_params(p), S, bucket_size=64), S)
    keep, near = ~m['prim'] & ~m['near'], m['near'] & ~m['prim']
    e, e_near = helpers.masked_rel_inf(got, ref, keep), helpers.masked_rel_inf(got, ref, near)
    worst, worst_near = max(worst, e), max(worst_near, e_near)
    if e >= 1e-4 or e_near >= 5e-2:
        bad.append((label, e, e_near))
    if (seed - a + 1) % 250 == 0:
        print(f'{seed - a + 1} configurations: worst outside the masks {worst:.2e}, behind a borderline pair {worst_near:.2e}, beyond the bars: {len(bad)}', flush=True)
print(f'{b - a} configurations: worst outside the masks {worst:.2e}, behind a borderline pair {worst_near:.2e}, beyond the bars: {len(bad)}')
for x in bad[:20]:
    print('  ', x)
