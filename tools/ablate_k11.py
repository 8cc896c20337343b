"""K11 timing experiments (fgs_debug_set_option key 7): full kernel vs no atomics vs no step loop, S2 and the layered scene; also the
distribution of live pixels per live bucket."""
import sys, statistics, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from harness.scenes import make_garden_like, orbit_views
from harness import trainer as T
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev = torch.device('cuda:0')
params = make_garden_like(3_000_000)
for off in (0.0, -3.0):
    p = {k: v.clone() for k, v in params.items()}; p['opacities'] = p['opacities'] + off
    g = T.Gaussians(p, dev)
    views = [v.to(dev) for v in orbit_views(8)]
    out = {}
    for rnd in range(4):
        v = views[rnd]
        S = T.extract_settings(v, 16, v.background_color)
        fw = be.forward(*g.tensors(), S)
        gi = torch.randn_like(fw.image) / fw.image.numel()
        args = (torch.empty(0, device=dev), gi, fw.image, g.means, g.scales, g.rotations, g.opacities, g.sh_coefficients_rest, fw.buffers, S, fw.state)
        for ab in (0, 1, 2, 4, 8):
            be.lib.fgs_debug_set_option(7, ab)
            be.backward(*args); torch.cuda.synchronize()
            be.profile_enable(True); be.profile_read()
            for _ in range(3): be.backward(*args)
            torch.cuda.synchronize(); t, c = be.profile_read()['blend_backward']; be.profile_enable(False)
            out.setdefault(ab, []).append(t / c)
        be.lib.fgs_debug_set_option(7, 0)
        if rnd == 0:
            n = g.means.shape[0]
            lay = be.blob_layout(1, n, 1920, 1080, fw.state[1], fw.state[2])
            npx = be.view(fw.buffers[1], lay, 'n_processed', torch.int32).view(-1, 192).long()
            mx = npx.max(dim=1).values
            nb = (mx + 63) // 64
            live_px = 0; live_b = int(nb.sum())
            for tb in range(int(nb.max())):
                sel = nb > tb
                live_px += int((npx[sel] > tb * 64).sum())
            print(f'opacity offset {off}: live buckets {live_b}, mean live pixels per live bucket {live_px / max(live_b, 1):.1f} of 192')
    for ab, name in ((0, 'full'), (1, 'no atomics'), (2, 'no step loop'), (4, 'no atomics of >1024-tile footprints'), (8, 'no atomics of >256-tile footprints')):
        print(f'  offset {off} {name:14s} median {statistics.median(out[ab]):.4f} ms  min {min(out[ab]):.4f}')
