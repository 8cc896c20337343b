"""Does the ORDER of the projected records matter to the renderer of the Gaussian-sharded step (the hypothesis DESIGN.md left
open)? On G ranks the renderer receives its view's records shard-major (all of shard 0, then shard 1, ...): Gaussians that are neighbours in the
scene's Morton order -- and therefore in the same tiles -- sit V / G records apart. One GPU, one process: the records of the whole scene in K1's
compaction order (~ memory = Morton order) are permuted into (a) that order itself (= what interleaving the shards slot = rank * G + shard would
restore), (b) shard-major order for G = 8, (c) a random order, and K2..K11 run on each: stage times of the blend kernels per order."""
import sys, statistics, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
import bench
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
sys.argv = ['bench.py']
params, views, _ = bench.build_scene(bench.parse())
dev = torch.device('cuda:0'); be = default_backend()
g = T.Gaussians(params, dev)
ORDER = ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')
P = {k: getattr(g, k).detach() for k in ORDER}
n = P['means'].shape[0]
rec = torch.empty((1, n, 56), dtype=torch.uint8, device=dev); cnt = torch.zeros((1, 2), dtype=torch.int32, device=dev)
G = 8
res_ms = {}
for vi in (0, 1, 3, 6):
    v = views[vi].to(dev); S = T.extract_settings(v, g.active_sh_bases, v.background_color)
    tgt = be.inference(*(P[k] for k in ORDER), S, True, True) * 0.9
    be.shard_preprocess(*(P[k] for k in ORDER), [S], rec, cnt)
    V, I = cnt[0].tolist()
    base = rec[0, :V].clone()
    idx = torch.arange(V, device=dev)
    major = torch.cat([idx[s::G] for s in range(G)])
    counts = [len(range(s, V, G)) for s in range(G)]
    orders = {'memory order (interleaved shards)': (idx, None),
              'shard-major, G = 8': (major, None),
              'shard-major, interleaved by the renderer': (major, counts),      # fgs_forward_from_shard_records: what ShardedTrainer.render does now
              'random': (torch.randperm(V, device=dev), None)}
    for name, (perm, sc) in orders.items():
        records = base[perm].contiguous()
        def cut():
            res = be.forward_from_records(records.reshape(-1), V, I, S, 15, shard_counts=sc)
            gl = be.l1_dssim(res.image, tgt, 0.8, 0.2, with_grad=True)[1]
            be.backward_to_records(gl, res.image, res.buffers, S, res.state, 15, shard_counts=sc)
        for _ in range(2): cut()
        torch.cuda.synchronize(); be.profile_enable(True); be.profile_read()
        for _ in range(4): cut()
        torch.cuda.synchronize(); pr = be.profile_read(); be.profile_enable(False)
        for k in ('shard_records', 'create_instances', 'blend_forward', 'blend_backward'):
            res_ms.setdefault((name, k), []).append(pr[k][0] / pr[k][1])
print(f'S2, views 0 1 3 6, records path (forward_from_records + backward_to_records), median stage ms per record order')
for name in ('memory order (interleaved shards)', 'shard-major, G = 8', 'shard-major, interleaved by the renderer', 'random'):
    print(f'  {name:42s} ' + '  '.join(f'{k} {statistics.median(res_ms[(name, k)]):.4f}' for k in ('shard_records', 'create_instances', 'blend_forward', 'blend_backward')))
