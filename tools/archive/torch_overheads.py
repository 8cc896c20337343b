"""Which framework-side kernels (fills, copies, elementwise passes) does one training iteration of the harness launch, and from where?
torch.profiler over a few iterations of bench.py's single-GPU step, grouped by kernel name with the Python call site."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from harness import trainer as T
from torch.profiler import profile, ProfilerActivity
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0')
g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in views]
targets = [torch.rand(3, v.height, v.width, device=dev) for v in views]
for i in range(3): T.training_iteration(g, views[i], targets[i], i)
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(N): T.training_iteration(g, views[i], targets[i], 3 + i)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or not e.name.startswith('aten::'): continue
    if e.name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::mul', 'aten::add_', 'aten::add', 'aten::zeros', 'aten::ones_like', 'aten::clone', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::empty_like'):
        dt = sum(k.duration for k in e.kernels) if e.kernels else 0.0
        if dt <= 0: continue
        stack = [s for s in (e.stack or []) if 'site-packages/torch' not in s and 'dist-packages/torch' not in s][:2]
        rows.append((e.name, tuple(e.input_shapes[0]) if e.input_shapes else (), dt, ' <- '.join(s.split('/')[-1] for s in stack)))
agg = {}
for name, shape, dt, where in rows:
    k = (name, shape, where); a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += dt
print(f'per iteration (of {N}): device time of framework-side ops that launch kernels')
for (name, shape, where), (n, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  {dt / N:7.1f} us  x{n / N:4.1f}  {name:16s} {str(shape):22s} {where}')
