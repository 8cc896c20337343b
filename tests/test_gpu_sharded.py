"""GPU tests (-m gpu) of the Gaussian-sharded multi-GPU path on ONE device: the cut pipeline (shard_preprocess ->
forward_from_records -> backward_to_records -> shard_backward) against the whole pipeline of the same library, at test and at
full (1080p, 1 M Gaussians) size, and the multi-owner trainer logic (LocalShardGroup == what G ranks compute) against the
replicated-parameter trainer. The RCCL exchanges themselves are covered by the world-size-2 gloo test (tests/test_sharded.py)
and run under `torch.distributed.run` by bench.py.

Tolerances: V and I exact (same K1 on the same Gaussians). The image differs from the whole pipeline's only through the order
of equal-depth Gaussians (record order instead of index order): 1e-6 (full size: <1e-4 of the pixels may exceed it). Gradients: float atomics in a different order, 1e-4 of
the tensor's max (relative inf-norm), as for the fused/unfused comparison in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import helpers
from harness.scenes import make_garden_like, make_s0, orbit_views
from test_sharded import ORDER, _run_sharded_by_hand

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LRS = {'means': 1.6e-4, 'sh_coefficients_0': 2.5e-3, 'sh_coefficients_rest': 1.25e-4, 'opacities': 2.5e-2, 'scales': 5e-3, 'rotations': 1e-3}


def _cut_vs_whole(be, params, RS, n_shards, grad_scale, strict=True):
    n = params['means'].shape[0]
    dp = {k: v.to(DEV).contiguous() for k, v in params.items()}
    whole = be.forward(*(dp[k] for k in ORDER), RS)
    gi = (torch.randn(3, RS.height, RS.width, generator=torch.Generator().manual_seed(7)) * grad_scale).to(DEV)
    info_ref = torch.zeros(2, n, device=DEV)
    ref = be.backward(info_ref, gi, whole.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'],
                      whole.buffers, RS, whole.state)
    dens = [torch.zeros(2, len(range(s, n, n_shards)), device=DEV) for s in range(n_shards)]
    rendered, table, grads = _run_sharded_by_hand(be, dp, [RS], [gi], n_shards, dens, dev=DEV)
    res = rendered[0][0]
    assert int(table[:, 0, 0].sum()) == whole.state[0] and int(table[:, 0, 1].sum()) == whole.state[1]
    assert res.state[1] == whole.state[1]
    if strict:
        assert (res.image - whole.image).abs().max().item() <= 1e-6
    else:       # at 1 M Gaussians a few thousand pairs share a 32-bit depth key; the rare overlapping pair blends in the other order
        assert helpers.rel_inf(res.image.cpu().numpy(), whole.image.cpu().numpy()) < 1e-4        # measured 2.9e-5
    for s in range(n_shards):
        for g, r, k in zip(grads[s], ref, ORDER):
            assert torch.isfinite(g).all(), k
            a, b = g.cpu().numpy(), r[s::n_shards].cpu().numpy()
            assert helpers.rel_inf(a, b) < 1e-4, (k, s, helpers.rel_inf(a, b))       # full size (1 M, 1080p, 8 shards): measured 3.7e-6
        assert helpers.rel_inf(dens[s].cpu().numpy(), info_ref[:, s::n_shards].cpu().numpy()) < 1e-4
    return table[:, 0]


@pytest.mark.parametrize('n_shards', [1, 2, 8])
def test_cut_pipeline_equals_whole_pipeline_s0(hip_backend, n_shards):
    params, view = make_s0(n=3000)
    _, RS = helpers.settings_pair(view, device=DEV)
    _cut_vs_whole(helpers.poisoned(hip_backend), params, RS, n_shards, 1e-2)


@pytest.mark.parametrize('K,aa', [(4, True), (9, False), (1, True)])
def test_cut_pipeline_sh_degrees_and_antialiasing(hip_backend, K, aa):
    """Proper antialiasing (opacity scaling in K1, its chain rule in K12) and lower active SH degrees through the records path."""
    params, view = make_s0(seed=4, n=2500)
    _, RS = helpers.settings_pair(view, K, aa, device=DEV)
    _cut_vs_whole(helpers.poisoned(hip_backend), params, RS, 3, 1e-2)


def test_cut_pipeline_with_huge_footprints(hip_backend):
    """Screen-filling Gaussians take K1's workgroup path and, on the sharded path, are spread through the record order by the
    pack kernel (they trade places with regular records): V, I, image and gradients must not notice."""
    from harness.scenes import View
    p, v = make_s0(seed=13, n=6000)
    v = View(v.w2c, v.position, 1280, 960, 1000.0, 1000.0, 640.0, 480.0, 0.2, 1e4, torch.zeros(3))
    p['scales'][:40] = p['scales'][:40] + 3.2            # > 1024 candidate tiles each
    p['opacities'][:40] -= 3.0                           # faint, so that everything behind them still receives gradients
    _, RS = helpers.settings_pair(v, device=DEV)
    for n_shards in (1, 3):
        table = _cut_vs_whole(helpers.poisoned(hip_backend), p, RS, n_shards, 1e-3, strict=False)
        assert int(table[:, 0].sum()) > 3000


def test_cut_pipeline_full_size(hip_backend):
    """BASELINE.json full size: 1 M garden-like Gaussians at 1920x1080, 8 shards; also: strided ownership balances the
    per-shard record counts (the all-to-all message sizes) to a few percent."""
    params = make_garden_like(1_000_000)
    v = orbit_views(8)[2]
    _, RS = helpers.settings_pair(v, device=DEV)
    table = _cut_vs_whole(hip_backend, params, RS, 8, 1.0 / (3 * 1080 * 1920), strict=False)
    per_shard = table[:, 0].double()
    assert (per_shard.max() - per_shard.min()) / per_shard.mean() < 0.05


def test_shard_backward_sums_views_on_device(hip_backend):
    """Three views through one K1 / K12 launch each: gradients == sum of the three whole-pipeline backward passes."""
    params = make_garden_like(30_000)
    params['scales'] = params['scales'] + 0.7
    dp = {k: v.to(DEV).contiguous() for k, v in params.items()}
    views = [helpers.settings_pair(v, device=DEV)[1] for v in orbit_views(8, width=640, height=360, focal=473.0)[:3]]
    gis = [(torch.randn(3, 360, 640, generator=torch.Generator().manual_seed(i)) / (3 * 360 * 640)).to(DEV) for i in range(3)]
    be = helpers.poisoned(hip_backend)
    _, _, grads = _run_sharded_by_hand(be, dp, views, gis, 2, dev=DEV)
    total = None
    for v, gi in zip(views, gis):
        whole = be.forward(*(dp[k] for k in ORDER), v)
        ref = be.backward(None, gi, whole.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'],
                          whole.buffers, v, whole.state)
        total = [r.clone() for r in ref] if total is None else [t + r for t, r in zip(total, ref)]
    for s in range(2):
        for g, t, k in zip(grads[s], total, ORDER):
            assert helpers.rel_inf(g.cpu().numpy(), t[s::2].cpu().numpy()) < 1e-4, (k, s)


@pytest.mark.parametrize('fused,steps', [(True, 3), (False, 3), (True, 1), (False, 1)])
def test_local_shard_group_equals_replicated_trainer(hip_backend, fused, steps):
    """4 owners x 4 views against ViewParallelTrainer summing the four per-view gradients itself: ONE step held to the strict 1e-4 max-norm bar
    (alpha-test flips cannot compound inside one step: a regression in record_of_slot / pack_acc that hits a few records fails here), and three steps
    with the drift allowance explained below (advisor finding on this file)."""
    from harness.distributed import SEGMENTS, ViewParallelTrainer
    from harness.sharded import LocalShardGroup
    params = {k: v.to(DEV) for k, v in make_garden_like(40_000).items()}
    params['scales'] = params['scales'] + 0.7
    views = orbit_views(8, width=640, height=360, focal=473.0)[:4]
    RS = [helpers.settings_pair(v, device=DEV)[1] for v in views]
    targets = [torch.rand(3, 360, 640, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(4)]
    grp = LocalShardGroup(hip_backend, params, LRS, 4, fused=fused)
    tr = ViewParallelTrainer(hip_backend, params, LRS)
    start = {k: params[k].clone() for k in SEGMENTS}
    # Both sides start from the same NON-ZERO Adam moments: from zero moments the first steps are lr * g / |g|, and an entry whose tiny
    # gradient changes sign under another summation order moves by 2 lr -- that would measure the optimizer's sensitivity, not the
    # agreement of the two gradient paths (the oracle tests of the fused path seed their moments for the same reason).
    for i, k in enumerate(SEGMENTS):
        m0, v0 = (t.to(DEV) for t in helpers.seeded_moments(params[k].shape, 11 + i))
        o, n, shape = tr.layout[k]
        tr.exp_avg[o:o + n].view(shape).copy_(m0)
        tr.exp_avg_sq[o:o + n].view(shape).copy_(v0)
        for s, t in enumerate(grp.ranks):
            o, n, shape = t.layout[k]
            t.exp_avg[o:o + n].view(shape).copy_(m0[s::4])
            t.exp_avg_sq[o:o + n].view(shape).copy_(v0[s::4])
    for _ in range(steps):
        grp.step(RS, targets)
        tr.step_count += 1
        total = torch.zeros_like(tr.grad_arena)
        for s, t in zip(RS, targets):
            tr._render_backward(s, lambda img: tr.image_gradient(img, t) * 0.25, True)
            total += tr.grad_arena
        tr.grad_arena.copy_(total)
        tr._adam(0, tr.param_arena.numel(), 0)
    got = grp.gather_parameters()
    for k in SEGMENTS:
        d_ref, d_got = (tr.params[k] - start[k]).cpu().numpy(), (got[k] - start[k]).cpu().numpy()
        assert np.abs(d_ref).max() > 0
        # Two correct pipelines whose gradients differ by summation order (1e-7, tools/archive/diag_shard_interleave.py) do not stay within 1e-4 entry by entry
        # over THREE steps: a parameter moved by 1e-7 flips an alpha >= 1/255 decision somewhere in the next render, and the Gaussians of that pixel then
        # differ at the 1e-4 level (which entries do depends on the noise: the interleaved record placement of round 4 moved the worst one from below to
        # above 1e-4). So: all but 1e-4 of the entries within 1e-4 of the largest update, none beyond 2e-3.
        helpers.log_note('sharded_vs_replicated', f'{helpers.rel_inf(d_got, d_ref):.3e}', tensor=k, steps=steps, fused=int(fused))
        if steps == 1:
            assert helpers.rel_inf(d_got, d_ref) < 1e-4, (k, helpers.rel_inf(d_got, d_ref))
            continue
        assert helpers.outlier_fraction(d_got, d_ref, 0.0, 1e-4 * float(np.abs(d_ref).max())) < 1e-4, (k, 'entries beyond 1e-4 of the largest update')
        assert helpers.rel_inf(d_got, d_ref) < 2e-3, (k, helpers.rel_inf(d_got, d_ref))
    info = torch.cat([t.densification_info for t in grp.ranks], dim=1)
    full_info = torch.empty_like(tr.densification_info)
    for s, t in enumerate(grp.ranks):
        full_info[:, s::4] = t.densification_info
    assert helpers.rel_inf(full_info.cpu().numpy(), tr.densification_info.cpu().numpy()) < 1e-4 and info.shape[1] == 40_000
