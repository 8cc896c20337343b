"""Analysis of K10's per-tile timeline (library built by tools/k10_timeline.sh build; FGS_HIP_LIBRARY points at it)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py'] + (['--ply', os.environ['FGS_PLY']] if os.environ.get('FGS_PLY') else [])     # FGS_PLY: a trained scene instead of S2
params, views, _ = bench.build_scene(bench.parse())
params['opacities'] = params['opacities'] + shift
dev = torch.device('cuda:0'); be = default_backend()
raw = C.CDLL(os.environ['FGS_HIP_LIBRARY'])
raw.fgs_debug_k10_timeline.argtypes = [C.c_void_p, C.c_uint, C.c_int]
g = T.Gaussians(params, dev)
v = views[2].to(dev)
S = T.extract_settings(v, g.active_sh_bases, v.background_color)
P = g.tensors()
for _ in range(2): be.forward(*P, S)
torch.cuda.synchronize()
assert raw.fgs_debug_k10_timeline(None, 0, 1) == 0
res = be.forward(*P, S); torch.cuda.synchronize()
n_tiles = ((v.width + 15) // 16) * ((v.height + 11) // 12)
buf = np.zeros(n_tiles * 4, np.uint64)
assert raw.fgs_debug_k10_timeline(buf.ctypes.data, n_tiles, 0) == 0
t = buf.reshape(n_tiles, 4); t = t[t[:, 1] > 0]
start, end = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
n_list, blk, xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64), (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64), (t[:, 3] & np.uint64(0xf)).astype(np.int64)
t0 = start.min(); span = end.max() - t0; dur = end - start
print(f'opacity shift {shift}: {len(t)} tiles, span {span / 100:.1f} us; tile duration us: median {np.median(dur) / 100:.1f}, p10 {np.percentile(dur, 10) / 100:.1f}, p90 {np.percentile(dur, 90) / 100:.1f}, '
      f'p99 {np.percentile(dur, 99) / 100:.1f}, max {dur.max() / 100:.1f}; list length median {np.median(n_list):.0f}, max {n_list.max()}')
print(f'  workgroups in flight, average over the span: {dur.sum() / span:.0f}')
edges = np.linspace(0, span, 21)
print('  mean workgroups in flight per 5 % slice:', [int((np.minimum(end - t0, b) - np.maximum(start - t0, a)).clip(min=0).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])])
order = np.sort(start - t0)
print('  start of workgroup #k at (share of span): 50%% %.3f, 90%% %.3f, 99%% %.3f, last %.3f' % tuple(order[[len(t) // 2, 9 * len(t) // 10, 99 * len(t) // 100, -1]] / span))
late = (end - t0) > 0.8 * span
print(f'  tiles still running in the last 20 % of the span: {int(late.sum())}; their duration us median {np.median(dur[late]) / 100:.1f}, started at (share of span) median {np.median((start[late] - t0) / span):.2f}; '
      f'correlation(duration, start) {np.corrcoef(dur, start - t0)[0, 1]:.2f}')
print('  end of the last tile per XCD (share of span):', [round(float((end[xcc == x].max() - t0) / span), 2) for x in range(8) if (xcc == x).any()])
print('  sum of tile durations per XCD (us):', [int(dur[xcc == x].sum() / 100) for x in range(8) if (xcc == x).any()])
