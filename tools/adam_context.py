"""Does the Adam kernel run slower inside the training iteration than alone? HIP events around optimizer.step() on S2: (a) in the iteration, (b) in the
iteration behind an idle gap of ~50 / ~200 us (torch.cuda._sleep), (c) alone, back to back on the same gradients, (d) alone behind the gap.
Printed: mean / min / max ms of the step() region per variant (the region is one launch, `adam_kernel<1, true>`)."""
import os, sys, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + '/faster-gaussian-splatting_amd']
import bench
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0')
g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in views]
tg = {id(v): T.render_image_benchmark(g, v).clone() * 0.9 for v in views}


def iteration(i, gap_cycles=0, keep_grads=False):
    v = views[i % 8]
    g.update_learning_rate(i + 1)
    image = T.render_image_training(g, v, update_densification_info=True, bg_color=v.background_color)
    loss = T.photometric_loss(image, tg[id(v)])
    loss.backward(gradient=T._unit_gradient(loss))
    if gap_cycles:
        torch.cuda._sleep(gap_cycles)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.optimizer.step(); e1.record()
    if not keep_grads:
        g.optimizer.zero_grad()
    return e0, e1


def alone(gap_cycles=0):
    if gap_cycles:
        torch.cuda._sleep(gap_cycles)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.optimizer.step(); e1.record()
    return e0, e1


def report(name, pairs):
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in pairs]
    print(f'{name:44s} mean {sum(ms) / len(ms):.4f}  min {min(ms):.4f}  max {max(ms):.4f}  ({len(ms)} launches)', flush=True)


for i in range(12): iteration(i)
torch.cuda.synchronize()
CYC = 2400                                           # ~1 us of _sleep at 2.4 GHz
for rnd in range(3):
    k = 100 * rnd
    report('in the iteration', [iteration(k + i) for i in range(24)])
    report('in the iteration, ~50 us idle in front', [iteration(k + 24 + i, 50 * CYC) for i in range(24)])
    report('in the iteration, ~200 us idle in front', [iteration(k + 48 + i, 200 * CYC) for i in range(24)])
    iteration(k + 72, keep_grads=True)
    report('alone, back to back', [alone() for _ in range(24)])
    report('alone, ~200 us idle in front', [alone(200 * CYC) for _ in range(24)])
    g.optimizer.zero_grad()
