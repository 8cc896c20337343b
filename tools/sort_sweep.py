"""Random-length sweep of the two device sorts against numpy's stable argsort (the checkers of tests/test_radix_sort.py): lengths around the
workgroup / wave / table boundaries and log-uniform random ones up to 20 M, the generic sort at the tile-key and depth-key bit ranges and the depth sort
(key - bits(near), 9-bit digits) over its pass counts. usage: python tools/sort_sweep.py [N_CASES] [SEED]"""
import os, sys
import numpy as np
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import test_radix_sort as T
from FasterGSCudaBackend._backend import default_backend
be = default_backend()
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
edges = [8192 * k + d for k in (1, 2, 3, 7, 244, 245) for d in (-1, 0, 1)] + [64 * k + d for k in (1, 127, 128) for d in (-1, 0, 1)] + [2, 3, 511, 513, 4096 * 512 - 1, 4096 * 512 + 1]
done = 0
for c in range(cases):
    n = int(edges[c]) if c < len(edges) else int(np.exp(rng.uniform(0, np.log(2e7 if c % 10 == 0 else 3e6))))
    T._check(be, n, np.uint32, 32, 'cuda', 1000 + c)
    T._check(be, n, np.uint16, int(rng.integers(1, 17)), 'cuda', 2000 + c, n_distinct=int(rng.integers(1, 20000)))
    T._check(be, n, np.uint32, int(rng.integers(17, 33)), 'cuda', 3000 + c, n_distinct=int(rng.integers(1, 200000)))
    near, far = T.DEPTH_RANGES[c % len(T.DEPTH_RANGES)]
    T._check_depth(be, n, near, far, 'cuda', 4000 + c, n_distinct=None if c % 3 else max(1, n // 7))
    done += 1
    if done % 25 == 0:
        print(f'{done} lengths, last n = {n}: ok', flush=True)
print(f'{done} lengths x 4 sorts: all equal to the stable argsort')
