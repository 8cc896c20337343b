"""Per-rank COMPUTE cost of the Gaussian-sharded multi-GPU step, measured on ONE MI355X.

LocalShardGroup runs everything G ranks would compute for one global batch of G views (each owner projects its shard for all
G views, each view is rendered once from the gathered records, each owner runs K12 for G views + Adam on its shard), with
local copies instead of RCCL. One rank's share is total / G. Printed next to the single-GPU iteration of the same scene and
the wire volume a rank would send per step, from which the xGMI time is estimated (links and rates: see DESIGN.md 6).

usage: python tools/sharded_emulation.py [--scene S2] [--worlds 2 4 8] [--steps 6]
"""
import argparse, json, os, sys, time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / 'faster-gaussian-splatting_amd')]
os.chdir(REPO)

import bench                                                              # scene construction shared with bench.py
from FasterGSCudaBackend._backend import default_backend
from harness import trainer as T
from harness.sharded import LocalShardGroup

ap = argparse.ArgumentParser()
ap.add_argument('--scene', default='S2')
ap.add_argument('--worlds', type=int, nargs='+', default=[2, 4, 8])
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--unfused', action='store_true', help='phase C as K12 + separate Adam launch')
a = ap.parse_args()

sys.argv = ['bench.py', '--scene', a.scene]
params, views, workload = bench.build_scene(bench.parse())
dev = torch.device('cuda:0')
be = default_backend()
g = T.Gaussians(params, dev)
g.training_setup(training_cameras_extent=5.0)
views = [v.to(dev) for v in views]
S = [T.extract_settings(v, g.active_sh_bases, v.background_color) for v in views]
with torch.no_grad():
    targets = [be.inference(*g.tensors(), s, True, True) * 0.9 for s in S]
lr = T.GARDEN_LR
lrs = {'means': lr['means_init'] * 5.0, **{k: lr[k] for k in T.PARAM_ORDER[1:]}}
full = {k: getattr(g, k).detach() for k in T.PARAM_ORDER}
out = {'workload': workload}

# single-GPU iteration (autograd path of bench.py)
for i in range(3):
    T.training_iteration(g, views[i % len(views)], targets[i % len(views)], i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    T.training_iteration(g, views[i % len(views)], targets[i % len(views)], i)
torch.cuda.synchronize()
single_ms = (time.perf_counter() - t0) / a.steps * 1e3
out['single_gpu_ms_per_iteration'] = single_ms

LINK_GBS = 76.8          # one xGMI link, one direction (153.6 GB/s bidirectional); every pair of GPUs has its own link
for G in a.worlds:
    grp = LocalShardGroup(be, full, lrs, G, fused=not a.unfused)
    batch = lambda i: [(i * G + r) % len(views) for r in range(G)]
    for i in range(2):
        idx = batch(i)
        grp.step([S[j] for j in idx], [targets[j] for j in idx])
    torch.cuda.synchronize()
    be.profile_enable(True); be.profile_read()
    t0 = time.perf_counter()
    for i in range(a.steps):
        idx = batch(i)
        grp.step([S[j] for j in idx], [targets[j] for j in idx])
    torch.cuda.synchronize()
    total_ms = (time.perf_counter() - t0) / a.steps * 1e3
    prof = be.profile_read(); be.profile_enable(False)
    tab = grp.last_counts.double()                                          # [shard, view, (V, I)]
    v_pair = float(tab[:, :, 0].mean())                                     # records of one (shard, view) pair
    # a rank sends (G-1) record messages and (G-1) accumulator messages, one per peer, each over that peer's own link
    wire_ms = (56.0 + 36.0) * v_pair / (LINK_GBS * 1e9) * 1e3
    sent_mb = (56.0 + 36.0) * v_pair * (G - 1) / 1e6
    replicated_mb = 2 * 236.0 * len(full['means']) * (G - 1) / G / 1e6
    per_rank = total_ms / G
    out[f'G{G}'] = {
        'compute_ms_per_rank_step': per_rank, 'stage_ms_per_rank_step': {k: v[0] / a.steps / G for k, v in prof.items() if v[1] > 0},
        'records_per_shard_view': v_pair, 'sent_MB_per_rank_step': sent_mb, 'replicated_zero1_MB_per_rank_step': replicated_mb,
        'xgmi_ms_ideal_per_link': wire_ms, 'est_step_ms': per_rank + wire_ms,
        'est_speedup_vs_single': G * single_ms / (per_rank + wire_ms),
        'zero1_est_step_ms': single_ms - 0.78 * (G - 1) / G + 2 * 236.0 * len(full['means']) / G / (LINK_GBS * 1e9) * 1e3,
    }
    del grp
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
