"""What could ANY scheme that streams K5's per-Gaussian inputs (a depth-sorted side array, a payload carried through the sort) win at most?
K5 (create_instances) and the offsets scan read one record / one count per visible Gaussian through the depth-sorted index list: a random gather.
Here the SAME scene is stored in the depth order of view 0 instead of Morton order, so that for view 0 the sorted list is ascending and both
gathers become streams -- with no extra pass, no extra bytes written by anybody: the floor of every such scheme. Everything else about the view is
unchanged (same Gaussians, same instances, same image). Prints the binning stages of view 0 for both storage orders, three alternating rounds."""
import os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py'] + sys.argv[1:]          # e.g. --scene S0 for a dry run on the simulation (tests/sim/run_with_sim.py)
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device(os.environ.get('FGS_TOOL_DEVICE', 'cuda:0')); be = default_backend()
v = views[0].to(dev)
depth = (params['means'].to(dev) @ v.w2c[2, :3] + v.w2c[2, 3])
order = torch.argsort(depth, stable=True).cpu()
by_depth = {k: p[order].contiguous() for k, p in params.items()}
stages = ['preprocess', 'depth_sort', 'offsets_scan', 'create_instances', 'tile_sort', 'extract_ranges', 'bucket_scan', 'blend_forward']
images = {}
for rnd in range(3):
    for tag, p in (('morton order', params), ('depth order of view 0', by_depth)):
        g = T.Gaussians(p, dev); g.training_setup(training_cameras_extent=5.0)
        target = T.render_image_benchmark(g, v).clone()
        images[tag] = target
        for i in range(3): T.training_iteration(g, v, target, i)
        be.profile_enable(True); be.profile_read()
        for i in range(10): T.training_iteration(g, v, target, 3 + i)
        torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        print(f'round {rnd}  {tag:22s} ' + '  '.join(f'{k} {pr[k][0] / 10:.4f}' for k in stages), flush=True)
        del g
a, b = images['morton order'], images['depth order of view 0']
print(f'images of the two storage orders differ by at most {float((a - b).abs().max()):.2e}')
