"""Prints the share of K1's wave-cycles per phase (library built by tools/k1_phase_timer.sh build; FGS_HIP_LIBRARY points at it)."""
import ctypes as C, os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
raw = C.CDLL(os.environ['FGS_HIP_LIBRARY'])
raw.fgs_debug_k1_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
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
    assert raw.fgs_debug_k1_phases(out, 1) == 0
    tot = float(sum(out[:6]))
    print(f'active SH bases {K}: preprocess stage {pr["preprocess"][0] / 8:.4f} ms per view (instrumented build); share of wave-cycles per phase:')
    for i, nm in enumerate(names): print(f'   {100.0 * out[i] / tot:5.1f} %   {nm}')
