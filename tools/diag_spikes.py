"""Where do the occasional slow blocks of the training loop come from? Per 10-iteration window: wall time and what the caching allocator did
(new segments = hipMalloc calls, freed segments, retries), on S2 through the harness' training_iteration (the bench's step)."""
import sys, time, gc, torch
sys.path[:0] = [__import__('os').environ['ROOT'], __import__('os').environ['ROOT'] + '/faster-gaussian-splatting_amd']
import bench
from harness import trainer as T
from FasterGSCudaBackend._backend import default_backend
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in views]
tg = {id(v): T.render_image_benchmark(g, v).clone() * 0.9 for v in views}
for i in range(5): T.training_iteration(g, views[i % 8], tg[id(views[i % 8])], i)
torch.cuda.synchronize()
def snap():
    s = torch.cuda.memory_stats()
    return s.get('segment.all.allocated', 0), s.get('segment.all.freed', 0), s.get('num_alloc_retries', 0), s.get('reserved_bytes.all.current', 0) / 1e9
prev = snap(); gc_count = [0]
gc.callbacks.append(lambda phase, info: gc_count.__setitem__(0, gc_count[0] + (phase == 'start' and info['generation'] == 2)))
for w in range(40):
    t0 = time.perf_counter()
    for i in range(10): T.training_iteration(g, views[(w * 10 + i) % 8], tg[id(views[(w * 10 + i) % 8])], 5 + w * 10 + i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    cur = snap()
    print(f'window {w:2d}: {ms:.3f} ms/it  new segments {cur[0] - prev[0]}  freed {cur[1] - prev[1]}  retries {cur[2] - prev[2]}  reserved {cur[3]:.2f} GB  gen2 gcs {gc_count[0]}', flush=True)
    prev = cur
