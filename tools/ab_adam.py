"""A/B of the Adam kernel's float4 pieces per thread inside ONE process (interleaved rounds, median of HIP-event times)."""
import sys, torch, statistics
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev = torch.device('cuda:0')
n = 3_000_000
sizes = [3 * n, 3 * n, 45 * n, n, 3 * n, 4 * n]
P = [torch.randn(s, device=dev) for s in sizes]; G = [torch.randn(s, device=dev) for s in sizes]
M = [torch.zeros(s, device=dev) for s in sizes]; V = [torch.zeros(s, device=dev) for s in sizes]
lrs = [1e-4] * 6
res = {1: [], 2: [], 4: [], 'nt': []}
for rnd in range(8):
    for u in (1, 2, 4, 'nt'):
        be.lib.fgs_debug_set_option(2, 1 if u == 'nt' else 0)
        be.lib.fgs_debug_set_option(1, 1 if u == 'nt' else u)
        be.adam_step_multi(G, P, M, V, [rnd + 1] * 6, lrs, 0.9, 0.999, 1e-15)
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        for _ in range(3): be.adam_step_multi(G, P, M, V, [rnd + 2] * 6, lrs, 0.9, 0.999, 1e-15)
        torch.cuda.synchronize(); t, c = be.profile_read()['adam']; be.profile_enable(False)
        res[u].append(t / c)
for u in res: print('unroll', u, 'median ms', round(statistics.median(res[u]), 4), 'min', round(min(res[u]), 4), 'TB/s', round(59 * n * 28 / statistics.median(res[u]) / 1e9, 3))
