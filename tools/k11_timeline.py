"""Analysis of K11's per-item timeline (library built by tools/k11_timeline.sh build; FGS_HIP_LIBRARY points at it)."""
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
raw.fgs_debug_k11_timeline.argtypes = [C.c_void_p, C.c_uint, C.c_int]
g = T.Gaussians(params, dev)
v = views[2].to(dev)
S = T.extract_settings(v, g.active_sh_bases, v.background_color)
P = g.tensors()
gi = torch.randn(3, v.height, v.width, device=dev) / (3 * v.height * v.width)
def run():
    res = be.forward(*P, S)
    be.backward(None, gi, res.image, P[0], P[1], P[2], P[3], P[5], res.buffers, S, res.state)
    torch.cuda.synchronize()
for _ in range(2): run()
assert raw.fgs_debug_k11_timeline(None, 0, 1) == 0
run()
N = 1 << 18
buf = np.zeros(N * 4, np.uint64)
assert raw.fgs_debug_k11_timeline(buf.ctypes.data, N, 0) == 0
t = buf.reshape(N, 4)
t = t[t[:, 1] > 0]
start, end = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)              # 100 MHz real-time counter: 10 ns ticks, chip-wide
steps, cycles = (t[:, 2] & np.uint64(0xffff)).astype(np.int64), (t[:, 2] >> np.uint64(16)).astype(np.float64)
xcc = (t[:, 3] & np.uint64(0xf)).astype(np.int64)
t0 = start.min(); span = end.max() - t0
dur = end - start
print(f'opacity shift {shift}: {len(t)} work items, span {span / 100:.1f} us; item duration us: median {np.median(dur) / 100:.1f}, p10 {np.percentile(dur, 10) / 100:.1f}, '
      f'p90 {np.percentile(dur, 90) / 100:.1f}, max {dur.max() / 100:.1f}; steps per item median {np.median(steps):.0f} (p10 {np.percentile(steps, 10):.0f}); '
      f'shader cycles per step median {np.median(cycles / np.maximum(steps, 1)):.0f}')
print(f'  items in flight, average over the span: {dur.sum() / span:.0f}')
edges = np.linspace(0, span, 21)
conc = [int((np.minimum(end - t0, b) - np.maximum(start - t0, a)).clip(min=0).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])]
print('  mean items in flight per 5 % slice of the span:', conc)
order = np.sort(start - t0)
print('  start of item #k at (share of span): k=0 %.3f, 10%% %.3f, 25%% %.3f, 50%% %.3f, 75%% %.3f, 90%% %.3f, last %.3f' % tuple(order[[0, len(t) // 10, len(t) // 4, len(t) // 2, 3 * len(t) // 4, 9 * len(t) // 10, -1]] / span))
print('  items per XCD:', np.bincount(xcc, minlength=8).tolist(), ' end of the last item per XCD (share of span):', [round(float((end[xcc == x].max() - t0) / span), 2) for x in range(8) if (xcc == x).any()])
early = dur[(start - t0) < 0.3 * span]; late = dur[(start - t0) > 0.7 * span]
print(f'  item duration us, started in the first 30 % of the span: median {np.median(early) / 100:.1f}; in the last 30 %: median {np.median(late) / 100 if len(late) else float("nan"):.1f}')
n_px = steps - 63
hist = np.bincount(np.clip(n_px // 24, 0, 8), minlength=9)
print('  live pixels per item (bins of 24, last = 192):', hist.tolist(), ' share of all steps spent on fill / drain (63 of n_px + 63):', round(float(63 * len(steps) / steps.sum()), 3))
