"""In-process A/B of K11's formulations: variant 3 (systolic, lane = Gaussian, compacted pixels) against variant 4 (lane = pixel walk, matrix-core
reduction) on S2, the layered scene (opacity logits - 3) and, with FGS_PLY, a trained export: blend_backward stage time per launch (HIP events,
interleaved rounds) and the largest relative difference of the six gradients between the two.
usage: [FGS_PLY=trained.ply] python tools/ab_k11m.py [variants, default 3,4]"""
import os, statistics, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from harness import trainer as T
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev = torch.device('cuda:0')
variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '3,4').split(',')]
if os.environ.get('FGS_ABLATE'):
    assert be.lib.fgs_debug_set_option(7, int(os.environ['FGS_ABLATE'])) == 0
if os.environ.get('FGS_K11M_BLOCKS'):
    assert be.lib.fgs_debug_set_option(13, int(os.environ['FGS_K11M_BLOCKS'])) == 0


def run(tag, params, views):
    g = T.Gaussians(params, dev)
    res = {v: [] for v in variants}
    worst = {}
    for rnd in range(6):
        v = views[rnd % len(views)].to(dev)
        S = T.extract_settings(v, g.active_sh_bases, v.background_color)
        fw = be.forward(*g.tensors(), S)
        gi = torch.randn_like(fw.image) / fw.image.numel()
        grads = {}
        for var in variants:
            be.lib.fgs_debug_set_backward_variant(var)
            args = (torch.empty(0, device=dev), gi, fw.image, g.means, g.scales, g.rotations, g.opacities, g.sh_coefficients_rest, fw.buffers, S, fw.state)
            grads[var] = [t.clone() for t in be.backward(*args)]
            torch.cuda.synchronize()
            be.profile_enable(True); be.profile_read()
            for _ in range(3):
                be.backward(*args)
            torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
            res[var].append(pr['blend_backward'][0] / pr['blend_backward'][1])
        if rnd == 0:       # what two runs of the SAME variant differ by (float atomics arrive in another order): the floor for the comparison below
            be.lib.fgs_debug_set_backward_variant(variants[0])
            again = [t.clone() for t in be.backward(torch.empty(0, device=dev), gi, fw.image, g.means, g.scales, g.rotations, g.opacities, g.sh_coefficients_rest, fw.buffers, S, fw.state)]
            for name, a, b in zip(('means', 'scales', 'rotations', 'opacities', 'sh0', 'sh_rest'), grads[variants[0]], again):
                worst[(f'{variants[0]} (second run)', name)] = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
        if rnd < 2 and len(variants) > 1:
            for var in variants[1:]:
                for name, a, b in zip(('means', 'scales', 'rotations', 'opacities', 'sh0', 'sh_rest'), grads[variants[0]], grads[var]):
                    d = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
                    worst[(var, name)] = max(worst.get((var, name), 0.0), d)
        del fw, grads
    be.lib.fgs_debug_set_backward_variant(3)
    print(tag)
    for var in variants:
        print(f'  variant {var}: blend_backward median {statistics.median(res[var]):.4f} ms  (min {min(res[var]):.4f}, max {max(res[var]):.4f})')
    for (var, name), d in worst.items():
        print(f'  variant {var} vs {variants[0]}: max |diff| / max |grad| of {name}: {d:.2e}')
    sys.stdout.flush()


sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
run('S2', params, views)
p2 = dict(params); p2['opacities'] = params['opacities'] - 3.0
run('layered (S2, opacity logits - 3)', p2, views)
del params, p2
if os.environ.get('FGS_PLY'):
    sys.argv = ['bench.py', '--ply', os.environ['FGS_PLY']]
    params, views, what = bench.build_scene(bench.parse())
    run(what, params, views)
