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
if os.environ.get('FGS_SCENE') == 'surface':          # bench.py's `surface_scene` extra: 2 M thin opaque disks on surfaces, look-at cameras
    import math
    from harness.scenes import look_at_view, make_surface_scene
    params = make_surface_scene(2_000_000)
    views = [look_at_view((6.4 * math.cos(2 * math.pi * k / 8), -(1.0 + 1.6 * (k % 3)), 6.4 * math.sin(2 * math.pi * k / 8)), (0.0, 1.3, 0.0), 1920, 1080, 1420.0) for k in range(8)]
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

# how far the lists are walked (max_n_processed of the training forward) against their length, for the tiles that set the span
lay = be.blob_layout(1, g.means.shape[0], v.width, v.height, res.state[1], res.state[2])
mx = be.view(res.buffers[1], lay, 'max_n_processed', torch.int32)[:n_tiles].cpu().numpy().astype(np.int64)
rng = be.view(res.buffers[1], lay, 'ranges', torch.int32).reshape(-1, 2)[:n_tiles].cpu().numpy().astype(np.int64)
ln = rng[:, 1] - rng[:, 0]
print(f'  lists: total instances {ln.sum()}, walked {mx.sum()} ({mx.sum() / max(ln.sum(), 1):.2f}); walked per tile median {np.median(mx):.0f}, p99 {np.percentile(mx, 99):.0f}, max {mx.max()}')
top = np.argsort(-dur)[:10]
tile_of = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
print('  ten longest tiles: duration us / list length:', [(round(float(dur[i]) / 100, 1), int(n_list[i])) for i in top])
print(f'  ns per walked Gaussian (sum of tile durations / walked instances): {dur.sum() * 10 / max(mx.sum(), 1):.1f}; tiles with lists >= 1024: {(ln >= 1024).sum()}, >= 2048: {(ln >= 2048).sum()}; walked >= 1024: {(mx >= 1024).sum()}')
