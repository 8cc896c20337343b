"""The Gaussian-sharded path on ONE device over the suite's fuzz configurations: shard_preprocess -> forward_from_records -> backward_to_records -> shard_backward with
1 .. 8 shards against the whole pipeline of the same library (tests/test_gpu_sharded._cut_vs_whole: counts exact, image 1e-4, every gradient and the densification
statistics 1e-4 of their maximum). usage: python tools/sharded_sweep.py A B"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers, test_gpu_sharded as TS
from FasterGSCudaBackend._backend import default_backend
be = helpers.poisoned(default_backend())
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    p, view, K, aa, label = helpers.fuzz_configuration(seed)
    n = p['means'].shape[0]
    _, RS = helpers.settings_pair(view, K, aa, device='cuda')
    shards = 1 + seed % 8
    if n < shards:
        shards = 1
    try:
        TS._cut_vs_whole(be, p, RS, shards, 1e-2, strict=False)
    except AssertionError as exc:
        bad.append((label, shards, str(exc)[:200]))
    if (seed - a + 1) % 100 == 0:
        print(f'{seed - a + 1} configurations: {len(bad)} beyond the bars', flush=True)
print(f'{b - a} configurations, 1 .. 8 shards: {len(bad)} beyond the bars')
for x in bad[:20]:
    print('  ', x)
