"""Random sweep of the device-side adaptive density control (fgs_adc_plan / fgs_adc_apply) against the numpy restatement of Model.py:312-366 that the tests use
(tests/test_densify.check_adc_against_restatement): random sizes across the scan's block boundaries (4096 Gaussians per block), random seeds, the three
option combinations. usage: python tools/adc_sweep.py [N_CASES] [SEED]"""
import os, sys
import numpy as np
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import test_densify as T
from oracle import oracle as O
from FasterGSCudaBackend._backend import default_backend
O.build()
be = default_backend()
cases, seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)
original = T._adc_case
edges = [4096 * k + d for k in (1, 2, 16) for d in (-1, 0, 1)]
for c in range(cases):
    n = int(edges[c]) if c < len(edges) else int(np.exp(rng.uniform(np.log(3000), np.log(400_000))))
    seed = int(rng.integers(0, 1 << 30))
    T._adc_case = lambda n=700, seed=seed, device='cpu', _s=seed: original(n=n, seed=_s, device=device)
    prune_large, with_state = [(True, True), (False, True), (True, False)][c % 3]
    counts = T.check_adc_against_restatement(be, O, 'cuda', n=n, prune_large=prune_large, with_state=with_state)
    if (c + 1) % 20 == 0:
        print(f'{c + 1} cases, last n = {n} seed {seed}: counts {counts} ok', flush=True)
print(f'{cases} cases: parameters, moments and counts equal to the restatement')
