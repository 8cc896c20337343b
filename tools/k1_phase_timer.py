"""Prints the share of K1's wave-cycles per phase (library built by tools/k1_phase_timer.sh build; FGS_HIP_LIBRARY points at it)."""
import ctypes as C, os, sys, torch
import numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
raw = C.CDLL(os.environ['FGS_HIP_LIBRARY'])
raw.fgs_debug_k1_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
raw.fgs_debug_k1_phase_waves.argtypes = [C.c_void_p, C.c_uint]
g = T.Gaussians(params, dev)
names = ['0 camera + mean load, depth cull', '1 opacity/scale/rotation loads, projection, bounds', '2 flattened exact tile count (<= 64 candidates)',
         '3 footprints > 64 candidates', '4 hot slots, SH colour, record write', '5 tile-count store, barrier, compaction atomic, key/index store']
for K in (16, 1):
    S = [T.extract_settings(v.to(dev), K, v.to(dev).background_color) for v in views]
    for s in S[:2]: be.inference(*g.tensors(), s, True, True)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    assert raw.fgs_debug_k1_phases(out, 1) == 0
    be.profile_enable(True); be.profile_read()
    for s in S: be.inference(*g.tensors(), s, True, True)
    torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
    n_waves = (g.means.shape[0] + 63) // 64
    table = np.zeros(n_waves * 8, np.uint64)
    assert raw.fgs_debug_k1_phase_waves(table.ctypes.data, n_waves) == 0
    assert raw.fgs_debug_k1_phases(out, 1) == 0
    tot = float(sum(out[:6]))
    print(f'active SH bases {K}: preprocess stage {pr["preprocess"][0] / 8:.4f} ms per view (instrumented build); share of wave-cycles per phase:')
    for i, nm in enumerate(names): print(f'   {100.0 * out[i] / tot:5.1f} %   {nm}')
    # is the kernel's time a tail? per-wave cycles summed over the 8 views (wave w holds Gaussians 64 w .. 64 w + 63 in every view)
    t = table.reshape(n_waves, 8)[:, :6].astype(np.float64) / 8.0
    total = t.sum(axis=1); active = total > 0
    pct = lambda a, q: float(np.percentile(a[active], q))
    print(f'   per-wave cycles per view (all phases): median {pct(total, 50):.0f}, p90 {pct(total, 90):.0f}, p99 {pct(total, 99):.0f}, p99.9 {pct(total, 99.9):.0f}, max {total.max():.0f}')
    for ph in (2, 3, 5):
        print(f'   phase {ph}: median {pct(t[:, ph], 50):.0f}, p99 {pct(t[:, ph], 99):.0f}, p99.9 {pct(t[:, ph], 99.9):.0f}, max {t[:, ph].max():.0f};  waves above 10 x the median total: {int((t[:, ph] > 10 * pct(total, 50)).sum())}')
    wg = total[: (n_waves // 4) * 4].reshape(-1, 4)      # a workgroup = 4 consecutive waves: its slowest wave holds the other three at the barrier
    print(f'   slowest wave of a workgroup / mean wave of the workgroup: median {float(np.median(wg.max(axis=1) / np.maximum(wg.mean(axis=1), 1))):.2f}')
