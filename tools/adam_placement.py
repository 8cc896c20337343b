"""Does the RELATIVE placement of Adam's four arrays (gradient, parameter, exp_avg, exp_avg_sq) change its bandwidth? One buffer, the four
arrays of one sh_rest-sized group (3 M x 45 floats = 540 MB each) carved at controlled offsets from a common 2 MB-aligned stride, the Adam
kernel timed for each padding; then the same with the arrays as separate torch allocations, several times (what the allocator happens to do)."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev = torch.device('cuda:0')
n = 3_000_000 * 45
nbytes = n * 4
def timed(g, p, m, v, reps=6):
    for _ in range(2): be.adam_step_multi([g], [p], [m], [v], [1], [1e-4], 0.9, 0.999, 1e-15)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); be.adam_step_multi([g], [p], [m], [v], [1], [1e-4], 0.9, 0.999, 1e-15); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    ts.sort(); return ts[len(ts) // 2]
stride0 = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
big = torch.empty(4 * stride0 + (64 << 20), dtype=torch.uint8, device=dev)
base = (-big.data_ptr()) % (2 << 20)
print(f'one group of {n} floats ({nbytes / 1e6:.0f} MB per array, 3.78 GB moved per step); median of 6 launches')
for pad in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19):
    arrs = []
    for k in range(4):
        off = base + k * (stride0 + pad)
        arrs.append(big[off:off + nbytes].view(torch.float32))
    arrs[0].normal_(); arrs[1].normal_(); arrs[2].zero_(); arrs[3].zero_()
    ms = timed(*arrs)
    print(f'  arrays {stride0 / 2**20:.0f} MiB + {pad:>8d} B apart: {ms:.4f} ms = {7 * nbytes / ms / 1e9:.2f} TB/s')
del big; torch.cuda.empty_cache()
for trial in range(4):
    junk = [torch.empty((trial * 37 + 1) << 20, dtype=torch.uint8, device=dev) for _ in range(trial)]
    g, p = torch.randn(n, device=dev), torch.randn(n, device=dev); m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ms = timed(g, p, m, v)
    offs = [(t.data_ptr() - g.data_ptr()) % (1 << 21) for t in (p, m, v)]
    print(f'  separate torch allocations, trial {trial}: {ms:.4f} ms = {7 * nbytes / ms / 1e9:.2f} TB/s   (offsets from g mod 2 MiB: {offs})')
    del g, p, m, v, junk; torch.cuda.empty_cache()
