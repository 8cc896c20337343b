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
res = {(1, 0): [], (2, 0): [], (4, 0): [], (1, 1): [], (2, 1): [], (4, 1): []}
for rnd in range(8):
    for (u, nt) in list(res):
        be.lib.fgs_debug_set_option(2, nt)
        be.lib.fgs_debug_set_option(1, u)
        be.adam_step_multi(G, P, M, V, [rnd + 1] * 6, lrs, 0.9, 0.999, 1e-15)
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        for _ in range(3): be.adam_step_multi(G, P, M, V, [rnd + 2] * 6, lrs, 0.9, 0.999, 1e-15)
        torch.cuda.synchronize(); t, c = be.profile_read()['adam']; be.profile_enable(False)
        res[(u, nt)].append(t / c)
for u in res: print('(float4 per thread, non-temporal)', u, 'median ms', round(statistics.median(res[u]), 4), 'min', round(min(res[u]), 4), 'TB/s', round(59 * n * 28 / statistics.median(res[u]) / 1e9, 3))
