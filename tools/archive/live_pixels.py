"""How many of a tile's 192 pixels are still live (not terminated) in each bucket K11 processes? Guides pixel compaction."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
offs = [float(x) for x in sys.argv[1:]] or [0.0]
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
import itertools
for off, vi in itertools.product(offs, (0, 1)):
    p2 = {k: v.clone() for k, v in params.items()}; p2['opacities'] = p2['opacities'] + off
    g = T.Gaussians(p2, dev); n = g.means.shape[0]
    v = views[vi].to(dev); S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    res = be.forward(*g.tensors(), S)
    lay = be.blob_layout(1, n, v.width, v.height, res.state[1], res.state[2])
    nt = ((v.width + 15) // 16) * ((v.height + 11) // 12)
    npr = be.view(res.buffers[1], lay, 'n_processed', torch.int32)[:nt * 192].view(nt, 192).long()
    ranges = be.view(res.buffers[1], lay, 'ranges', torch.int32)[:2 * nt].view(nt, 2).long()
    length = ranges[:, 1] - ranges[:, 0]
    mx = npr.max(dim=1).values
    nb_proc = (mx + 63) // 64                      # buckets K11 runs for the tile (those in front of max_n_processed)
    nb_all = (length + 63) // 64
    tot_b = int(nb_proc.sum()); live_sum = 0; hist = torch.zeros(7, dtype=torch.long, device=dev)
    maxb = int(nb_proc.max())
    for b in range(maxb):
        sel = nb_proc > b
        live = (npr[sel] > b * 64).sum(dim=1)      # pixels that still blend something at or after this bucket
        live_sum += int(live.sum())
        hist += torch.histc(live.float(), bins=7, min=0, max=192.001).long()
    print(f'opacity offset {off} view {vi}: buckets total {int(nb_all.sum())}, processed {tot_b}, mean live pixels per processed bucket {live_sum / tot_b:.1f} / 192')
    print('   live-pixel histogram (bins of ~27):', hist.tolist())
    steps_now = tot_b * 255; steps_compact = live_sum + 63 * tot_b
    print(f'   systolic steps now {steps_now / 1e6:.1f} M, with live-pixel compaction {steps_compact / 1e6:.1f} M ({100 * steps_compact / steps_now:.0f} %)')
