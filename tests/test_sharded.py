"""CPU tests of the Gaussian-sharded multi-GPU path (include/fgs_hip.h "Gaussian-sharded multi-GPU path",
harness/sharded.py) on the simulation backend:
 * the four cut-pipeline entry points, driven by hand for 2 shards in one process, reproduce fgs_forward / fgs_backward;
 * world_size-2 `gloo` processes running ShardedTrainer end with the parameters a single process gets from the summed
   two-view gradient (the same reference as tests/test_distributed.py)."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from test_distributed import LRS, _scene, _setup_paths, _single_process_reference

ORDER = ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')


def _run_sharded_by_hand(be, params, views, grad_images, n_shards, dens=None, dev='cpu'):
    """All of `views` through shard_preprocess -> forward_from_records -> backward_to_records -> shard_backward, with the two
    exchanges done by slicing. Returns per view (result, acc) and per shard the six gradients summed over the views."""
    K = params['sh_coefficients_rest'].shape[1]
    shards = [{k: v[s::n_shards].contiguous() for k, v in params.items()} for s in range(n_shards)]
    prim, recs, counts = [], [], []
    for sh in shards:
        n = sh['means'].shape[0]
        rec = torch.zeros((len(views), max(n, 1), 56), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((len(views), 2), dtype=torch.int32, device=dev)
        prim.append(be.shard_preprocess(*(sh[k] for k in ORDER), views, rec, cnt))
        recs.append(rec)
        counts.append(cnt)
    table = torch.stack(counts).cpu()                                   # [shard, view, (V, I)]
    rendered = []
    for v, view in enumerate(views):
        records = torch.cat([recs[s][v, :int(table[s, v, 0])] for s in range(n_shards)]).contiguous()
        res = be.forward_from_records(records.view(-1), records.shape[0], int(table[:, v, 1].sum()), view, K)
        acc = be.backward_to_records(grad_images[v], res.image, res.buffers, view, res.state, K)
        rendered.append((res, acc))
    grads = []
    for s, sh in enumerate(shards):
        sent = [int(table[s, v, 0]) for v in range(len(views))]
        pieces = []
        for v in range(len(views)):
            o = int(table[:s, v, 0].sum())
            pieces.append(rendered[v][1][o:o + sent[v]])
        out = tuple(torch.full_like(sh[k], float('nan')) for k in ORDER)     # every element must be written
        be.shard_backward(torch.cat(pieces).contiguous(), sent, prim[s], None if dens is None else dens[s], sh['means'], sh['scales'],
                          sh['rotations'], sh['opacities'], sh['sh_coefficients_rest'], views, out)
        grads.append(out)
    return rendered, table, grads


def _cut_equals_whole(params, s, n_shards):
    be = helpers.poisoned(helpers.sim_backend())      # scratch buffers start as NaN / 0xFF garbage
    whole = be.forward(*(params[k] for k in ORDER), s)
    torch.manual_seed(3)
    grad_image = torch.randn(3, s.height, s.width) * 1e-2
    n = params['means'].shape[0]
    info_ref = torch.zeros(2, n)
    ref = be.backward(info_ref, grad_image, whole.image, params['means'], params['scales'], params['rotations'], params['opacities'],
                      params['sh_coefficients_rest'], whole.buffers, s, whole.state)
    dens = [torch.zeros(2, len(range(sh, n, n_shards))) for sh in range(n_shards)]
    rendered, table, grads = _run_sharded_by_hand(be, params, [s], [grad_image], n_shards, dens)
    res = rendered[0][0]
    assert int(table[:, 0, 0].sum()) == whole.state[0] and int(table[:, 0, 1].sum()) == whole.state[1]        # V and I: exact
    assert res.state[1] == whole.state[1]
    assert torch.allclose(res.image, whole.image, rtol=0, atol=1e-6)
    for sh in range(n_shards):
        for g, r, k in zip(grads[sh], ref, ORDER):
            assert torch.isfinite(g).all(), k
            assert torch.allclose(g, r[sh::n_shards], rtol=1e-4, atol=1e-7), (k, (g - r[sh::n_shards]).abs().max())
        assert torch.allclose(dens[sh], info_ref[:, sh::n_shards], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('n_shards', [1, 2, 3])
def test_cut_pipeline_equals_whole_pipeline(n_shards):
    params, settings, _ = _scene()
    _cut_equals_whole(params, settings[0], n_shards)


@pytest.mark.parametrize('seed', [1, 4, 6, 9, 13, 22])
def test_cut_pipeline_equals_whole_pipeline_on_fuzz_configurations(seed):
    """The seeded configurations of the fuzz tests (helpers.fuzz_configuration: 2 Gaussians -- an empty shard --, counts around the wavefront
    size, 3000 Gaussians, odd image sizes, near / far planes, both antialiasing modes, culled / degenerate / screen-filling Gaussians) cut
    into three shards: records out, rendered from records, accumulators back, gradients on the owners == the undivided pipeline."""
    p, view, K, aa, label = helpers.fuzz_configuration(seed)
    _cut_equals_whole(p, helpers.settings_pair(view, K, aa)[1], 3)


@pytest.mark.parametrize('n_views', [2, 9])
def test_shard_backward_sums_over_views(n_views):
    """K12's in-register sum over the views of a launch (and the accumulate path across launches of 8 views, n_views = 9)."""
    params, settings, _ = _scene()
    be = helpers.poisoned(helpers.sim_backend())
    torch.manual_seed(4)
    views = [settings[i % 2] for i in range(n_views)]
    gis = [torch.randn(3, v.height, v.width) * 1e-2 for v in views]
    n = params['means'].shape[0]
    dens = [torch.zeros(2, n)]
    _, _, grads = _run_sharded_by_hand(be, params, views, gis, 1, dens)
    total, info_ref = None, torch.zeros(2, n)
    for v, gi in zip(views, gis):
        whole = be.forward(*(params[k] for k in ORDER), v)
        ref = be.backward(info_ref, gi, whole.image, params['means'], params['scales'], params['rotations'], params['opacities'],
                          params['sh_coefficients_rest'], whole.buffers, v, whole.state)
        total = [r.clone() for r in ref] if total is None else [t + r for t, r in zip(total, ref)]
    for g, t, k in zip(grads[0], total, ORDER):
        assert torch.allclose(g, t, rtol=1e-4, atol=1e-7), (k, (g - t).abs().max())
    assert torch.allclose(dens[0], info_ref, rtol=1e-5, atol=1e-8)


def test_empty_shard_and_no_visible():
    params, settings, _ = _scene()
    be = helpers.sim_backend()
    s = settings[0]
    empty = {k: v[:0].contiguous() for k, v in params.items()}
    cnt = torch.full((2, 2), 7, dtype=torch.int32)
    be.shard_preprocess(*(empty[k] for k in ORDER), [s, s], torch.zeros((2, 1, 56), dtype=torch.uint8), cnt)
    assert cnt.tolist() == [[0, 0], [0, 0]]
    res = be.forward_from_records(torch.zeros(0, dtype=torch.uint8), 0, 0, s, 15)
    assert torch.allclose(res.image, s.bg_color.view(3, 1, 1).expand_as(res.image))
    acc = be.backward_to_records(torch.ones_like(res.image), res.image, res.buffers, s, res.state, 15)
    assert acc.shape == (0, 9)
    # a shard none of whose Gaussians is visible still writes all its gradients (zeros)
    far = {k: v[:20].clone() for k, v in params.items()}
    far['means'][:, 2] = -50.0
    cnt = torch.zeros((1, 2), dtype=torch.int32)
    prim = be.shard_preprocess(*(far[k] for k in ORDER), [s], torch.zeros((1, 20, 56), dtype=torch.uint8), cnt)
    assert cnt.tolist() == [[0, 0]]
    out = tuple(torch.full_like(far[k], float('nan')) for k in ORDER)
    be.shard_backward(torch.zeros((0, 9)), [0], prim, None, far['means'], far['scales'], far['rotations'], far['opacities'],
                      far['sh_coefficients_rest'], [s], out)
    assert all(bool((g == 0).all()) for g in out)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _setup_paths()
        import helpers as h
        from harness.sharded import ShardedTrainer, shard_of
        params, settings, targets = _scene()
        tr = ShardedTrainer(h.sim_backend(), shard_of(params, rank, world), LRS)
        for _ in range(2):
            tr.step(settings, targets[rank])
        full = tr.gather_parameters()
        torch.save({'shard': {k: v.clone() for k, v in tr.params.items()}, 'full': full, 'info': tr.densification_info.clone(),
                    'counts': tr.last_counts}, Path(out_dir) / f'sharded_{rank}.pt')
    finally:
        dist.destroy_process_group()


def test_sharded_world2_gloo(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f'sharded_{i}.pt') for i in range(2)]
    ref_params, ref_info = _single_process_reference()
    start = _scene()[0]
    for k in ref_params:
        assert torch.equal(r[0]['full'][k], r[1]['full'][k]), k
        assert torch.allclose(r[0]['full'][k], ref_params[k], rtol=0, atol=1e-6), (k, (r[0]['full'][k] - ref_params[k]).abs().max())
        for i in range(2):
            assert torch.equal(r[i]['shard'][k], r[0]['full'][k][i::2]), k
        assert (r[0]['full'][k] - start[k]).abs().max() > 0
    for i in range(2):      # owners accumulate the densification statistics of BOTH views: no collective needed
        assert torch.allclose(r[i]['info'], ref_info[:, i::2], rtol=1e-5, atol=1e-7)
    assert torch.equal(r[0]['counts'], r[1]['counts']) and int(r[0]['counts'][..., 0].sum()) > 0


@pytest.mark.parametrize('world', [3])
def test_local_shard_group_matches_summed_gradient_reference(world):
    """G owners in one process (the twin that GPU tests and tools/sharded_emulation.py use): same update as the reference."""
    from harness.sharded import LocalShardGroup
    params, settings, targets = _scene()
    views = [settings[i % 2] for i in range(world)]
    tg = [targets[i % 2] for i in range(world)]
    grp = LocalShardGroup(helpers.sim_backend(), params, LRS, world)
    for _ in range(2):
        grp.step(views, tg)
    got = grp.gather_parameters()
    # reference: replicated parameters, gradients of the `world` views summed (each scaled 1/world), one Adam step
    from harness.distributed import SEGMENTS, ViewParallelTrainer
    tr = ViewParallelTrainer(helpers.sim_backend(), params, LRS)
    for _ in range(2):
        tr.step_count += 1
        total = torch.zeros_like(tr.grad_arena)
        for s, t in zip(views, tg):
            tr._render_backward(s, lambda img: tr.image_gradient(img, t) * (1.0 / world), False)
            total += tr.grad_arena
        tr.grad_arena.copy_(total)
        tr._adam(0, tr.param_arena.numel(), 0)
    for k in SEGMENTS:
        assert torch.allclose(got[k], tr.params[k], rtol=0, atol=1e-6), (k, (got[k] - tr.params[k]).abs().max())


def test_fused_and_unfused_phase_c_agree():
    """fgs_shard_backward_adam_fused == fgs_shard_backward + Adam (2 owners x 2 views, 2 steps; moments compared as well)."""
    from harness.sharded import LocalShardGroup
    params, settings, targets = _scene()
    views, tg = [settings[i % 2] for i in range(2)], [targets[i % 2] for i in range(2)]
    groups = [LocalShardGroup(helpers.sim_backend(), params, LRS, 2, fused=f) for f in (True, False)]
    for grp in groups:
        for _ in range(2):
            grp.step(views, tg)
    for a, b in zip(groups[0].ranks, groups[1].ranks):
        assert torch.allclose(a.param_arena, b.param_arena, rtol=0, atol=1e-6)
        assert torch.allclose(a.exp_avg, b.exp_avg, rtol=1e-5, atol=1e-9) and torch.allclose(a.exp_avg_sq, b.exp_avg_sq, rtol=1e-5, atol=1e-12)
        assert torch.equal(a.densification_info, b.densification_info)


def _worker_n(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _setup_paths()
        import helpers as h
        from harness.sharded import ShardedTrainer, shard_of
        params, settings, targets = _scene()
        tr = ShardedTrainer(h.sim_backend(), shard_of(params, rank, world), LRS)
        for _ in range(2):
            tr.step([settings[i % 2] for i in range(world)], targets[rank % 2])
        full = tr.gather_parameters()
        if rank == 0:
            torch.save(full, Path(out_dir) / 'full.pt')
    finally:
        dist.destroy_process_group()


def test_sharded_world4_gloo_equals_local_group(tmp_path):
    """Four processes exchanging through all_to_all_single (uneven splits, 4 x 4 count table) == the four owners stepped in one
    process with local copies: the collectives move exactly the slices the local twin takes."""
    from harness.sharded import LocalShardGroup
    mp.spawn(_worker_n, args=(4, 33500 + (os.getpid() % 2000), str(tmp_path)), nprocs=4, join=True)
    got = torch.load(tmp_path / 'full.pt')
    params, settings, targets = _scene()
    grp = LocalShardGroup(helpers.sim_backend(), params, LRS, 4)
    for _ in range(2):
        grp.step([settings[i % 2] for i in range(4)], [targets[i % 2] for i in range(4)])
    ref = grp.gather_parameters()
    for k in ref:
        assert torch.equal(got[k], ref[k]), k


def _fuzz_step_inputs(seed, world):
    import helpers as h
    p, view, K, aa, label = h.fuzz_configuration(seed)
    s = h.settings_pair(view, K, aa)[1]
    gen = torch.Generator().manual_seed(seed)
    targets = [torch.rand(3, view.height, view.width, generator=gen) for _ in range(world)]
    return p, [s] * world, targets


def _worker_fuzz(rank, world, port, out_dir, seed):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _setup_paths()
        import helpers as h
        from harness.sharded import ShardedTrainer, shard_of
        params, settings, targets = _fuzz_step_inputs(seed, world)
        tr = ShardedTrainer(h.sim_backend(), shard_of(params, rank, world), LRS)
        for _ in range(2):
            tr.step(settings, targets[rank])
        full = tr.gather_parameters()
        if rank == 0:
            torch.save(full, Path(out_dir) / 'full.pt')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('seed', [6, 22])
def test_sharded_world3_gloo_on_fuzz_configurations(tmp_path, seed):
    """Three processes over gloo on two of the fuzz configurations (127 and 64 Gaussians: shards of 43 / 42 / 42 and 22 / 21 / 21, narrow odd-sized
    images, 1 and 9 active SH bases): uneven all-to-all splits in both directions == the three owners stepped in one process."""
    from harness.sharded import LocalShardGroup
    mp.spawn(_worker_fuzz, args=(3, 35500 + (os.getpid() % 2000) + seed, str(tmp_path), seed), nprocs=3, join=True)
    got = torch.load(tmp_path / 'full.pt')
    params, settings, targets = _fuzz_step_inputs(seed, 3)
    grp = LocalShardGroup(helpers.sim_backend(), params, LRS, 3)
    for _ in range(2):
        grp.step(settings, targets)
    ref = grp.gather_parameters()
    for k in ref:
        assert torch.equal(got[k], ref[k]), k


@pytest.mark.parametrize('sh_bases', [1, 4])
def test_sharded_step_with_fewer_sh_bands(sh_bases):
    """SH degree 0 / 1 models through the sharded path (generic-R kernels, empty sh_rest): fused and unfused phase C agree."""
    from harness.scenes import View, make_s0
    from harness.sharded import LocalShardGroup
    params, v0 = make_s0(seed=9, n=150, sh_bases=sh_bases)
    views = []
    for shift in (0.0, 0.4):
        w2c = v0.w2c.clone()
        w2c[0, 3] = shift
        views.append(helpers.settings_pair(View(w2c, torch.tensor([-shift, 0.0, -4.0]), 48, 36, 48.0, 48.0, 24.0, 18.0, 0.2, 1e4, torch.zeros(3)),
                                           active_sh_bases=sh_bases)[1])
    targets = [torch.full((3, 36, 48), 0.4), torch.full((3, 36, 48), 0.6)]
    groups = [LocalShardGroup(helpers.sim_backend(), params, LRS, 2, fused=f) for f in (True, False)]
    for grp in groups:
        grp.step(views, targets)
    for a, b in zip(groups[0].ranks, groups[1].ranks):
        assert torch.allclose(a.param_arena, b.param_arena, rtol=0, atol=1e-6)
        assert torch.isfinite(a.param_arena).all()


def test_owner_side_densification_between_steps():
    """Owners densify their shards independently (harness.densify on `as_gaussians()`), adopt the result (`rebuild_from`) and keep
    training: sizes change per shard, Adam moments of the survivors are carried over, new Gaussians start from zero moments."""
    from harness import densify as D
    from harness.sharded import LocalShardGroup
    params, settings, targets = _scene()
    grp = LocalShardGroup(helpers.sim_backend(), params, LRS, 2)
    grp.step(settings, targets)
    sizes = []
    for t in grp.ranks:
        g = t.as_gaussians()
        assert float(g.densification_info[0].max()) >= 1.0                     # the owner saw the statistics of both views
        n0 = g.means.shape[0]
        kept_moment = g.optimizer.state[g.means]['exp_avg'].clone()
        stats = D.adaptive_density_control(g, 1e-7, 0.005, False, generator=torch.Generator().manual_seed(3), ops_backend=helpers.sim_backend())
        assert stats['cloned'] + stats['split'] > 0
        D.reset_densification_info(g)
        t.rebuild_from(g)
        sizes.append((n0, t.n))
        assert t.params['means'].shape[0] == t.n == g.means.shape[0] and t.densification_info.shape == (2, t.n)
        o, n, shape = t.layout['means']
        assert torch.equal(t.exp_avg[o:o + n].view(shape), g.optimizer.state[g.means]['exp_avg']) and kept_moment.abs().max() > 0
    assert any(a != b for a, b in sizes)
    images = grp.step(settings, targets)                                        # the next step runs on the new shards
    assert all(torch.isfinite(im).all() for im in images) and all(torch.isfinite(t.param_arena).all() for t in grp.ranks)
    # export after densification: shard sizes no longer follow the strided pattern -> shards are concatenated in rank order
    full = grp.gather_parameters()
    assert full['means'].shape[0] == sum(t.n for t in grp.ranks)
    assert torch.equal(full['means'][:grp.ranks[0].n], grp.ranks[0].params['means'])
    assert torch.equal(full['sh_coefficients_rest'][grp.ranks[0].n:], grp.ranks[1].params['sh_coefficients_rest'])
    # the shard's maintenance view carries the scene extent / SH cap given to the trainer
    t = grp.ranks[0]
    t.extent, t.max_sh_degree = 5.0, 2
    g2 = t.as_gaussians()
    assert g2.max_sh_degree == 2 and abs(g2.optimizer.param_groups[0]['lr'] - t.lrs['means']) < 1e-12


@pytest.mark.parametrize('n_shards', [2, 3, 5])
def test_interleaved_record_placement_changes_nothing_but_the_order(n_shards):
    """fgs_forward_from_shard_records / fgs_backward_to_shard_records: the renderer places the shards' records interleaved (Morton neighbourhood of
    strided owners restored: K11 0.51 -> 0.43 ms at S2 with 8 shards, profiles/r04_ab_sharded_order.txt). With UNEQUAL segment lengths the image must
    be bit-identical to the as-received placement (the visible list and therefore every tile's blending order are unchanged -- as long as no two
    visible Gaussians share a depth key: tied keys sort in primitive-slot order, which the interleaving changes; this scene has no ties, asserted below) and every accumulator
    record must come back at the position its record came in."""
    params, settings, _ = _scene()
    s = settings[0]
    be = helpers.poisoned(helpers.sim_backend())
    K = params['sh_coefficients_rest'].shape[1]
    n = params['means'].shape[0]
    cuts = sorted({0, n} | {int(n * f) for f in ([0.2, 0.55, 0.6, 0.9][:n_shards - 1])})      # contiguous, deliberately unequal shards
    recs, cnts = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        sh = {k: v[a:b].contiguous() for k, v in params.items()}
        rec = torch.zeros((1, max(b - a, 1), 56), dtype=torch.uint8)
        cnt = torch.zeros((1, 2), dtype=torch.int32)
        be.shard_preprocess(*(sh[k] for k in ORDER), [s], rec, cnt)
        recs.append(rec[0, :int(cnt[0, 0])])
        cnts.append(cnt[0].tolist())
    counts = [c[0] for c in cnts]
    assert len(set(counts)) > 1                                   # the interesting case
    records = torch.cat(recs).contiguous()
    V, I = sum(counts), sum(c[1] for c in cnts)
    depth_keys = records.numpy().view(np.uint32).reshape(-1, 14)[:, 12]        # a splat record = the 48-byte projected record + depth key + tile count
    assert len(np.unique(depth_keys)) == V                       # the premise of the bit-identical image: no tied depth keys
    torch.manual_seed(3)
    grad_image = torch.randn(3, s.height, s.width) * 1e-2
    plain = be.forward_from_records(records.view(-1), V, I, s, K)
    acc_plain = be.backward_to_records(grad_image, plain.image, plain.buffers, s, plain.state, K)
    inter = be.forward_from_records(records.view(-1), V, I, s, K, shard_counts=counts)
    acc_inter = be.backward_to_records(grad_image, inter.image, inter.buffers, s, inter.state, K, shard_counts=counts)
    assert torch.equal(inter.image, plain.image) and inter.state[:2] == plain.state[:2]
    assert torch.allclose(acc_inter, acc_plain, rtol=1e-5, atol=1e-9), (acc_inter - acc_plain).abs().max()
    with pytest.raises(RuntimeError):                             # counts that do not add up are refused
        be.forward_from_records(records.view(-1), V, I, s, K, shard_counts=[counts[0] + 1] + counts[1:])
