"""One whole training iteration (fgs_forward_async + fused loss + fgs_backward + one-launch Adam) captured into a HIP graph at S2: time per
replay next to the same calls issued eagerly (synchronous and synchronisation-free forward)."""
import sys, time, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd', '/root/repo/tests']
import helpers
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
from harness.scenes import make_garden_like, orbit_views
from test_gpu_graph import ORDER, _iteration
be = default_backend(); dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
params = make_garden_like(n)
v = orbit_views(8)[0].to(dev)
RS = T.extract_settings(v, 16, v.background_color)
P = {k: params[k].to(dev).clone() for k in ORDER}
M = {k: torch.zeros_like(P[k]) for k in ORDER}; V = {k: torch.zeros_like(P[k]) for k in ORDER}
target = be.inference(*[P[k] for k in helpers.NAMES], RS, True, True) * 0.9
sync = be.forward(*[P[k] for k in helpers.NAMES], RS)
cap = int(1.25 * sync.state[1]) + 4096
del sync


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


print('eager, fgs_forward (one host read)      ms/iteration', round(timed(lambda: _iteration(be, P, M, V, RS, target, None, 1)), 4))
print('eager, fgs_forward_async                ms/iteration', round(timed(lambda: _iteration(be, P, M, V, RS, target, cap, 1)), 4))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    _iteration(be, P, M, V, RS, target, cap, 1)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = _iteration(be, P, M, V, RS, target, cap, 1)
print('hipGraph replay of the captured iteration ms/iteration', round(timed(g.replay), 4))
