"""Time of one adaptive-density-control call / Morton re-ordering / prune at S2 size (3 M Gaussians, SH degree 3, with Adam moments):
device passes of csrc/densify.hip vs the torch-op formulation of the reference (harness.densify on the same device tensors)."""
import sys, time, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from harness import densify as D
from harness import trainer as T
from harness.scenes import make_garden_like
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
params = make_garden_like(n)


def fresh():
    g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
    for group in g.optimizer.param_groups:
        p = group['params'][0]
        g.optimizer.state[p] = {'step': 5, 'exp_avg': torch.randn_like(p) * 1e-3, 'exp_avg_sq': torch.rand_like(p) * 1e-6}
    info = torch.zeros(2, n, device=dev); info[0] = 10.0
    info[1] = torch.rand(n, device=dev) * 3e-3            # ~1/3 above the 2e-4 * 10 threshold
    g.densification_info = info
    return g


# Device passes only: the torch-op formulation this tool used to time beside them (10 ms / 250-370 ms / - at 3 M Gaussians on the same GPU,
# profiles/archive/r02_time_densify.txt) left the product in round 6.
times = {}
for rep in range(3):
    g = fresh(); torch.cuda.synchronize()
    t0 = time.perf_counter(); stats = D.adaptive_density_control(g, 2e-4, 0.005, True); torch.cuda.synchronize(); t1 = time.perf_counter()
    D.reset_densification_info(g)
    D.apply_morton_ordering(g); torch.cuda.synchronize(); t2 = time.perf_counter()
    mask = torch.rand(g.means.shape[0], device=dev) < 0.2; torch.cuda.synchronize(); t3 = time.perf_counter()
    D.prune(g, mask); torch.cuda.synchronize(); t4 = time.perf_counter()
    times.setdefault('adc', []).append(t1 - t0); times.setdefault('morton', []).append(t2 - t1); times.setdefault('prune', []).append(t4 - t3)
    del g
print(f'device passes N={n}: adaptive_density_control {min(times["adc"]) * 1e3:8.2f} ms   morton re-order {min(times["morton"]) * 1e3:8.2f} ms   prune 20% {min(times["prune"]) * 1e3:8.2f} ms   ({stats})')
