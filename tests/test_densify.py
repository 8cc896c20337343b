"""Adaptive density control / pruning / sorting with Adam-state surgery (reference Model.py:262-366).
CPU tests of the tensor semantics; the GPU end-to-end run is tests/test_gpu_training.py."""
import math

import numpy as np
import pytest

import torch

import helpers  # noqa: F401  (sets up import paths)
from harness import densify as D
from harness.scenes import make_s0
from harness import trainer as T
from harness.trainer import PARAM_ORDER, Gaussians


def _gaussians(n=200):
    params, _ = make_s0(n=n)
    g = Gaussians(params, 'cpu')
    g.training_setup(training_cameras_extent=4.0)
    for group in g.optimizer.param_groups:           # give every parameter a recognisable optimizer state
        p = group['params'][0]
        g.optimizer.state[p] = {'step': 7, 'exp_avg': torch.full_like(p, 0.5), 'exp_avg_sq': torch.full_like(p, 0.25)}
    return g


def test_adaptive_density_control_clone_split_prune(sim_backend):
    g = _gaussians()
    n = g.means.shape[0]
    info = torch.zeros(2, n)
    info[0] = 10.0
    info[1, :30] = 10.0 * 1e-3            # mean gradient 1e-3 >= 2e-4: densify the first 30
    g.densification_info = info
    with torch.no_grad():
        g.scales[:10] = math.log(0.5)     # large (> percent_dense * extent = 0.04): split
        g.scales[10:30] = math.log(0.01)  # small: clone
        g.opacities[100:105] = -10.0      # nearly transparent: pruned
        g.rotations[110] = 0.0            # degenerate quaternion: pruned
    before = {k: getattr(g, k).detach().clone() for k in PARAM_ORDER}
    stats = D.adaptive_density_control(g, 2e-4, 0.005, False, generator=torch.Generator().manual_seed(0), ops_backend=sim_backend)
    assert stats['cloned'] == 20 and stats['split'] == 10
    assert g.means.shape[0] == n + 20 + 2 * 10 - 10 - 5 - 1 == stats['total']
    # clones are exact copies; split children are shrunk by 1/1.6 and displaced
    kept_original = n - 10 - 5 - 1
    clones = g.means[kept_original:kept_original + 20]
    assert torch.equal(clones, before['means'][10:30])
    children = g.scales[kept_original + 20:]
    assert torch.allclose(children.exp(), torch.full_like(children, 0.5 * 0.625))
    # Adam state: survivors keep theirs, new entries start at zero, step is preserved
    for group in g.optimizer.param_groups:
        st = g.optimizer.state[group['params'][0]]
        assert st['step'] == 7 and st['exp_avg'].shape == group['params'][0].shape
        assert torch.all(st['exp_avg'][:kept_original] == 0.5) and torch.all(st['exp_avg'][kept_original:] == 0.0)
        assert torch.all(st['exp_avg_sq'][:kept_original] == 0.25) and torch.all(st['exp_avg_sq'][kept_original:] == 0.0)
    assert g.densification_info is None
    D.reset_densification_info(g)
    assert g.densification_info.shape == (2, g.means.shape[0])


def test_opacity_reset_and_morton_sort():
    g = _gaussians(64)
    D.reset_opacities(g)
    assert float(g.opacities.max()) <= -4.595119953155518 + 1e-6
    op_state = g.optimizer.state[g.opacities]
    assert torch.all(op_state['exp_avg'] == 0) and torch.all(g.optimizer.state[g.means]['exp_avg'] == 0.5)
    means_before = g.means.detach().clone()
    tag = torch.arange(64, dtype=torch.float32)
    g.optimizer.state[g.means]['exp_avg'][:, 0] = tag
    D.apply_morton_ordering(g)
    perm = g.optimizer.state[g.means]['exp_avg'][:, 0].long()
    assert sorted(perm.tolist()) == list(range(64)) and torch.equal(g.means.detach(), means_before[perm])


def test_callback_schedule_matches_trainer():
    fired = {'densify': [], 'morton': [], 'reset': [], 'sh': []}
    g = _gaussians(32)
    g.active_sh_degree = 0
    orig = (D.adaptive_density_control, D.apply_morton_ordering, D.reset_opacities)
    D.adaptive_density_control = lambda *a, **k: fired['densify'].append(cur) or {'total': 0}
    D.apply_morton_ordering = lambda g_, *a: fired['morton'].append(cur)
    D.reset_opacities = lambda g_: fired['reset'].append(cur)
    try:
        for cur in range(0, 16_001, 100):
            deg = g.active_sh_degree
            D.run_callbacks(g, cur)
            if g.active_sh_degree != deg:
                fired['sh'].append(cur)
    finally:
        D.adaptive_density_control, D.apply_morton_ordering, D.reset_opacities = orig
    assert fired['densify'][0] == 600 and fired['densify'][-1] == 14_900 and len(fired['densify']) == 144   # Trainer.py:120
    assert fired['morton'] == [0, 5_000, 10_000, 15_000] and fired['reset'] == [3_000, 6_000, 9_000, 12_000]   # :141,:156
    assert fired['sh'] == [1_000, 2_000, 3_000]                                                               # :114


def test_ply_round_trip(tmp_path):
    """Model.py:511-542 layout: channel-major SH, raw opacity / scale, normalised quaternion; save -> load is lossless."""
    from harness import ply
    g = _gaussians(37)
    d = ply.as_ply_dict(g)
    v = d['vertex']
    assert v.dtype.names[:6] == ('x', 'y', 'z', 'f_dc_0', 'f_dc_1', 'f_dc_2') and v.dtype.names[-4:] == ('rot_0', 'rot_1', 'rot_2', 'rot_3')
    assert len(v.dtype.names) == 3 + 3 + 45 + 1 + 3 + 4
    # f_rest is channel-major: f_rest_0..14 = red coefficients of bases 1..15
    assert torch.allclose(torch.from_numpy(v['f_rest_1'].copy()), g.sh_coefficients_rest[:, 1, 0]) \
        and torch.allclose(torch.from_numpy(v['f_rest_15'].copy()), g.sh_coefficients_rest[:, 0, 1])
    ply.save_ply(g, tmp_path / 'g.ply')
    back = ply.load_ply(tmp_path / 'g.ply')
    for k in PARAM_ORDER:
        ref = getattr(g, k).detach()
        if k == 'rotations':
            ref = ref / ref.norm(dim=1, keepdim=True)
        assert back[k].shape == ref.shape and torch.allclose(back[k], ref, atol=1e-7), k


def test_initialize_from_point_cloud():
    """Model.py:202-231: log RMS-3NN scales, identity quaternions, logit(0.1) opacities, sh0 = (rgb - 0.5) / C0."""
    from harness.scenes import initialize_from_point_cloud, root_mean_squared_knn_distances
    gen = torch.Generator().manual_seed(0)
    pts = torch.rand(500, 3, generator=gen)
    col = torch.rand(500, 3, generator=gen)
    p = initialize_from_point_cloud(pts, col)
    d = torch.cdist(pts, pts)
    d.fill_diagonal_(float('inf'))
    ref = d.square().topk(3, dim=1, largest=False).values.mean(dim=1).sqrt()
    assert torch.allclose(root_mean_squared_knn_distances(pts, chunk=128), ref, rtol=1e-4, atol=1e-7)   # cdist expands |a-b|^2: ~1e-5
    assert torch.allclose(p['scales'], ref.log()[:, None].expand(-1, 3), atol=1e-4)
    assert torch.equal(p['rotations'], torch.tensor([1.0, 0, 0, 0]).expand(500, 4))
    assert torch.allclose(torch.sigmoid(p['opacities']), torch.full((500, 1), 0.1))
    assert torch.allclose(0.5 + 0.28209479177387814 * p['sh_coefficients_0'][:, 0], col, atol=1e-6)
    assert p['sh_coefficients_rest'].shape == (500, 15, 3) and not p['sh_coefficients_rest'].any()
    m = initialize_from_point_cloud(pts, None, use_mcmc=True)
    assert torch.allclose(m['scales'], (0.1 * ref).log()[:, None].expand(-1, 3), atol=1e-4)
    assert torch.allclose(torch.sigmoid(m['opacities']), torch.full((500, 1), 0.5)) and torch.allclose(m['sh_coefficients_0'], torch.zeros(500, 1, 3))
    g = Gaussians(p, 'cpu')          # feeds the trainer unchanged
    assert g.means.shape == (500, 3)


def _sim_ops():
    be = helpers.sim_backend()

    def add_noise(raw_scales, raw_rotations, raw_opacities, means, current_lr):
        be.add_noise(raw_scales.contiguous(), raw_rotations.contiguous(), raw_opacities.contiguous(), torch.randn_like(means), means, current_lr)
    return be.relocation_adjustment, add_noise


def test_mcmc_densification_relocates_and_grows():
    """Model.py:367-457 on the simulation backend's relocation kernel: dead Gaussians become copies of sampled live ones with
    the shared opacity / scale of 3DGS-MCMC Eq. 9, the sampled ones lose their Adam moments, the set grows by 5 % up to the cap."""
    g = _gaussians(200)
    with torch.no_grad():
        g.opacities[:20] = -12.0                 # dead (sigmoid << min_opacity)
        g.rotations[20:25] = 0.0                 # degenerate quaternion: dead as well
        g.opacities[25:] = g.opacities[25:].clamp_min(-1.0)
    before = {k: getattr(g, k).detach().clone() for k in PARAM_ORDER}
    stats = D.mcmc_densification(g, 0.005, 200, generator=torch.Generator().manual_seed(0), ops=_sim_ops())    # cap reached: relocation only
    assert stats['relocated'] == 25 and stats['added'] == 0 and g.means.shape[0] == 200
    op = torch.sigmoid(g.opacities.detach()).flatten()
    assert float(op.min()) >= 0.005 - 1e-6                                                    # nothing dead is left
    # every relocated Gaussian sits exactly on a previously live one and shares its new opacity and scale
    for i in range(25):
        src = torch.where((before['means'][25:] == g.means.detach()[i]).all(dim=1))[0]
        assert src.numel() >= 1
        j = 25 + int(src[0])
        assert torch.equal(g.opacities.detach()[i], g.opacities.detach()[j]) and torch.equal(g.scales.detach()[i], g.scales.detach()[j])
        assert float(torch.sigmoid(g.opacities.detach()[j])) < float(torch.sigmoid(before['opacities'][j])) + 1e-7   # shared => not larger
        st = g.optimizer.state[g.means]
        assert torch.all(st['exp_avg'][j] == 0.0)
    assert g.densification_info is None
    # growth: 5 % more Gaussians, capped; the copies start with zero moments
    stats = D.mcmc_densification(g, 0.005, 208, generator=torch.Generator().manual_seed(1), ops=_sim_ops())
    assert stats['relocated'] == 0 and stats['added'] == 8 and g.means.shape[0] == 208        # min(cap, int(1.05 * 200)) = 208
    st = g.optimizer.state[g.means]
    assert st['exp_avg'].shape[0] == 208 and torch.all(st['exp_avg'][200:] == 0.0)
    assert D.mcmc_densification(g, 0.005, 208, generator=torch.Generator().manual_seed(2), ops=_sim_ops())['added'] == 0


def test_importance_pruning_and_noise_hook():
    g = _gaussians(100)
    scores = torch.arange(100, dtype=torch.float32)
    tag = g.means.detach().clone()
    removed = D.importance_pruning(g, scores, 0.3)                      # Model.py:465-470: k = int(0.3 * 99) + 1 = 30 lowest
    assert removed == 30 and g.means.shape[0] == 70 and torch.equal(g.means.detach(), tag[30:])
    assert g.optimizer.state[g.means]['exp_avg'].shape[0] == 70
    before = g.means.detach().clone()
    torch.manual_seed(0)
    D.post_optimizer_step(g, False, 1.6e-4, ops=_sim_ops())
    assert torch.equal(g.means.detach(), before)
    D.post_optimizer_step(g, True, 1.6e-4, ops=_sim_ops())             # Model.py:472-475: SGLD noise scaled by (1 - opacity) sigmoid gate
    moved = (g.means.detach() - before).abs().max()
    assert 0.0 < float(moved) < 5.0


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """Adam moments, step counts, learning rates and the SH degree survive save -> load: two more optimizer steps after a resume
    equal two more steps of the uninterrupted run (FusedAdam replaced by the same update in torch for this CPU test)."""
    from harness.checkpoint import load_checkpoint, save_checkpoint

    def adam_steps(g, start, count):            # the FusedAdam update rule (adam.cu:22-33) on CPU tensors, deterministic "gradients"
        for it in range(start, start + count):
            g.update_learning_rate(it + 1)
            for group in g.optimizer.param_groups:
                p = group['params'][0]
                grad = torch.sin(p.detach() * (it + 1))
                st = g.optimizer.state.setdefault(p, {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)})
                st['step'] += 1
                st['exp_avg'].mul_(0.9).add_(grad, alpha=0.1)
                st['exp_avg_sq'].mul_(0.999).addcmul_(grad, grad, value=0.001)
                bc1, bc2 = 1 - 0.9 ** st['step'], 1 - 0.999 ** st['step']
                p.data.addcdiv_(st['exp_avg'], (st['exp_avg_sq'] / bc2).sqrt() + 1e-15, value=-group['lr'] / bc1)

    a = _gaussians(50)
    a.optimizer.state.clear()
    a.active_sh_degree = 2
    adam_steps(a, 0, 3)
    save_checkpoint(a, tmp_path / 'ckpt.pt', iteration=3)
    b, it = load_checkpoint(tmp_path / 'ckpt.pt', 'cpu')
    assert it == 3 and b.active_sh_degree == 2 and b.means.shape == a.means.shape
    adam_steps(a, 3, 2)
    adam_steps(b, 3, 2)
    for k in PARAM_ORDER:
        assert torch.equal(getattr(a, k).detach(), getattr(b, k).detach()), k
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        sa, sb = a.optimizer.state[ga['params'][0]], b.optimizer.state[gb['params'][0]]
        assert sa['step'] == sb['step'] == 5 and torch.equal(sa['exp_avg'], sb['exp_avg']) and ga['lr'] == gb['lr']


# ---- the device passes of csrc/densify.hip (through the CPU simulation of the library) against the numpy restatement of Model.py ----
ORDER = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')


def _adc_case(n=700, seed=4, device='cpu'):
    """A scene in which every branch of Model.py:312-366 fires: clones, splits, dead opacities, degenerate rotations, oversized Gaussians."""
    params, _ = make_s0(seed=seed, n=n)
    gen = torch.Generator().manual_seed(seed)
    params['scales'][: n // 3] += 2.0                                   # large ones: split candidates / too large after the opacity reset
    params['scales'][n // 3: n // 3 + 20] += 3.2                        # oversized (pruned when prune_large_gaussians)
    params['opacities'][5:n:17] = -7.0                                  # dead
    params['rotations'][9:n:41] = 0.0                                   # degenerate
    info = torch.zeros(2, n)
    info[0] = torch.randint(0, 30, (n,), generator=gen).float()
    info[1] = torch.rand(n, generator=gen) * 6e-4 * info[0].clamp_min(1.0)      # mean gradients around the 2e-4 threshold
    m = {k: torch.randn(params[k].shape, generator=gen) * 1e-3 for k in ORDER}
    v = {k: torch.rand(params[k].shape, generator=gen) * 1e-6 for k in ORDER}
    return {k: params[k].to(device) for k in ORDER}, {k: m[k].to(device) for k in ORDER}, {k: v[k].to(device) for k in ORDER}, info.to(device)


def check_adc_against_restatement(be, oracle, device, n=700, prune_large=True, with_state=True, atol=0.0):
    P, M, V, info = _adc_case(n=n, device=device)
    noise_holder = {}

    def noise_fn(rows):
        noise_holder['z'] = torch.randn((rows, 3), generator=torch.Generator().manual_seed(11))
        return noise_holder['z'].to(device)
    out_p, out_m, out_v, counts = be.adaptive_density_control(info, [P[k] for k in ORDER], [M[k] for k in ORDER] if with_state else None,
                                                              [V[k] for k in ORDER] if with_state else None, 2e-4, 0.005, prune_large, 0.01, 5.0, noise_fn)
    np_of = lambda d: {k: d[k].cpu().numpy() for k in ORDER}
    ref_p, ref_m, ref_v, ref_counts = oracle.adaptive_density_control(np_of(P), np_of(M) if with_state else None, np_of(V) if with_state else None,
                                                                      info.cpu().numpy(), noise_holder['z'].numpy(), 2e-4, 0.005, prune_large, 0.01, 5.0)
    assert counts == ref_counts and min(counts) > 0, (counts, ref_counts)
    for i, k in enumerate(ORDER):
        a = out_p[i].cpu().numpy()
        assert a.shape == ref_p[k].shape, k
        if k in ('means', 'scales'):      # children: exp / log / sqrt of different libms; everything else is a copy
            assert np.abs(a - ref_p[k]).max() <= 2e-6 * (1.0 + np.abs(ref_p[k]).max()) + atol, k
        else:
            assert np.array_equal(a, ref_p[k]), k
        if with_state:
            assert np.array_equal(out_m[i].cpu().numpy(), ref_m[k]) and np.array_equal(out_v[i].cpu().numpy(), ref_v[k]), k
    return counts


@pytest.mark.parametrize('prune_large,with_state', [(True, True), (False, True), (True, False)])
def test_device_adaptive_density_control_matches_model_py(sim_backend, oracle, prune_large, with_state):
    check_adc_against_restatement(sim_backend, oracle, 'cpu', prune_large=prune_large, with_state=with_state)


def test_device_adaptive_density_control_across_scan_blocks(sim_backend, oracle):
    """The 4-way exclusive scan of csrc/densify.hip works in 4096-Gaussian workgroup blocks: two whole blocks and a ragged third."""
    check_adc_against_restatement(sim_backend, oracle, 'cpu', n=2 * 4096 + 37)


def test_device_gather_and_morton_order(sim_backend, oracle):
    P, M, V, _ = _adc_case(n=500)
    order = sim_backend.morton_order(P['means'])
    assert np.array_equal(order.numpy(), oracle.morton_order(P['means'].numpy()))                  # Model.py:459-463
    idx = torch.randperm(500, generator=torch.Generator().manual_seed(1))[:321]
    tensors = [P[k] for k in ORDER] + [M[k] for k in ORDER] + [V[k] for k in ORDER] + [P['means']]   # 19 tensors: two launches
    outs = sim_backend.gather_rows(tensors, idx)
    for t, o in zip(tensors, outs):
        assert torch.equal(o, t[idx])


def test_harness_densify_uses_the_device_passes(sim_backend, oracle):
    """harness.densify on top of the device passes: the Gaussians and moments the numpy restatement of Model.py:312-366 (oracle/oracle.py) gives,
    step counts kept; the gather launch behind prune / Morton order equals plain indexing. Without a backend there is nothing to fall back to."""
    params, _ = make_s0(seed=2, n=300)
    info = torch.zeros(2, 300); info[0] = 10.0; info[1, :60] = 10.0 * 1e-3
    make = lambda: T.Gaussians({k: v.clone() for k, v in params.items()}, 'cpu')
    ga, gb = make(), make()
    for g in (ga, gb):
        g.training_setup(training_cameras_extent=5.0)
        for group in g.optimizer.param_groups:
            p = group['params'][0]
            g.optimizer.state[p] = {'step': 3, 'exp_avg': torch.full_like(p, 0.5), 'exp_avg_sq': torch.full_like(p, 0.25)}
        g.densification_info = info.clone()
    with pytest.raises(RuntimeError, match='no backend'):
        D.adaptive_density_control(ga, 2e-4, 0.005, True, generator=torch.Generator().manual_seed(5))
    before = {k: getattr(gb, k).detach().numpy().copy() for k in ORDER}
    moments = {k: np.full(before[k].shape, 0.5, np.float32) for k in ORDER}, {k: np.full(before[k].shape, 0.25, np.float32) for k in ORDER}
    sb = D.adaptive_density_control(gb, 2e-4, 0.005, True, generator=torch.Generator().manual_seed(5), ops_backend=sim_backend)
    noise = torch.randn((2 * sb['split'], 3), generator=torch.Generator().manual_seed(5)).numpy()
    ref_p, ref_m, ref_v, ref_counts = oracle.adaptive_density_control(before, moments[0], moments[1], info.numpy(), noise, 2e-4, 0.005, True, 0.01, 5.0)
    assert (sb['kept'], sb['cloned'], sb['children_per_copy'], sb['split']) == tuple(ref_counts) and sb['total'] == ref_p['means'].shape[0] and gb.densification_info is None
    for k in ORDER:
        assert np.allclose(getattr(gb, k).detach().numpy(), ref_p[k], rtol=0, atol=2e-6), k
        stb = gb.optimizer.state[getattr(gb, k)]
        assert np.array_equal(stb['exp_avg'].numpy(), ref_m[k]) and np.array_equal(stb['exp_avg_sq'].numpy(), ref_v[k]) and stb['step'] == 3
    # re-ordering and pruning: the gather launch against plain indexing, from bit-identical sets
    ga = make()
    ga.training_setup(training_cameras_extent=5.0)
    ga.densification_info = None
    with torch.no_grad():
        for k in ORDER:
            getattr(ga, k).data = getattr(gb, k).detach().clone()
    for group in ga.optimizer.param_groups:
        p = group['params'][0]
        ga.optimizer.state[p] = {key: (val.clone() if torch.is_tensor(val) else val) for key, val in gb.optimizer.state[getattr(gb, group['name'])].items()}
    D.apply_morton_ordering(ga)
    D.apply_morton_ordering(gb, ops_backend=sim_backend)
    assert torch.equal(ga.means, gb.means) and torch.equal(ga.optimizer.state[ga.means]['exp_avg'], gb.optimizer.state[gb.means]['exp_avg'])
    mask = torch.zeros(ga.means.shape[0], dtype=torch.bool); mask[::3] = True
    D.prune(ga, mask); D.prune(gb, mask, ops_backend=sim_backend)
    assert torch.equal(ga.sh_coefficients_rest, gb.sh_coefficients_rest)


@pytest.mark.parametrize('device_passes', [False, True])
def test_partial_optimizer_state_keeps_moments_row_aligned(sim_backend, device_passes):
    """Round-2 advisor finding: with optimizer state on only SOME groups (a group that never received a gradient, a checkpoint saved
    without one group's moments) the device passes used to rebind every group without moments, leaving the groups that had state with
    moments of the OLD row count -- the next Adam step would read past them. Now the missing groups get the state Adam creates lazily
    (zero moments) and every group goes through the same gather / scatter; the plain-indexing prune for CPU tensors leaves stateless groups stateless."""
    g = _gaussians(150)
    n = g.means.shape[0]
    opt = g.optimizer
    for grp in opt.param_groups:
        if grp['name'] in ('rotations', 'sh_coefficients_rest'):
            del opt.state[grp['params'][0]]
    info = torch.zeros(2, n)
    info[0] = 10.0
    info[1, :20] = 10.0 * 1e-3
    g.densification_info = info
    with torch.no_grad():
        g.scales[:20] = math.log(0.01)
        g.opacities[50:60] = -10.0
    be = sim_backend if device_passes else None
    stats = D.adaptive_density_control(g, 2e-4, 0.005, False, generator=torch.Generator().manual_seed(0), ops_backend=sim_backend)
    assert g.means.shape[0] == stats['total'] == n + 20 - 10
    D.prune(g, torch.arange(g.means.shape[0]) % 3 == 0, ops_backend=be)
    for grp in g.optimizer.param_groups:
        p = grp['params'][0]
        st = g.optimizer.state.get(p)
        if grp['name'] in ('rotations', 'sh_coefficients_rest'):
            assert not st or (st['exp_avg'].shape == p.shape and float(st['exp_avg'].abs().max()) == 0.0)
        else:
            assert st['exp_avg'].shape == p.shape and st['exp_avg_sq'].shape == p.shape, grp['name']
            assert float(st['exp_avg'][0].abs().max()) == 0.5        # survivors keep their moments


def test_mcmc_regularisation_gradients_equal_autograd():
    """Loss.py:17-18 with Model.py:136-142 under USE_MCMC: lambda * mean(sigmoid(opacities)) + lambda * mean(exp(scales)); the harness adds the analytic
    gradient between backward and optimizer step (harness.densify.add_mcmc_regularisation_gradients)."""
    g = _gaussians(120)
    torch.manual_seed(5)
    photometric = {k: torch.randn_like(getattr(g, k)) * 1e-3 for k in ('opacities', 'scales')}
    for k, v in photometric.items():
        getattr(g, k).grad = v.clone()
    D.add_mcmc_regularisation_gradients(g, 0.01, 0.02)
    o = g.opacities.detach().clone().requires_grad_(True)
    s = g.scales.detach().clone().requires_grad_(True)
    (0.01 * torch.sigmoid(o).mean() + 0.02 * torch.exp(s).mean()).backward()
    assert torch.allclose(g.opacities.grad, photometric['opacities'] + o.grad, rtol=1e-6, atol=1e-10)
    assert torch.allclose(g.scales.grad, photometric['scales'] + s.grad, rtol=1e-6, atol=1e-10)
    before = g.opacities.grad.clone()
    D.add_mcmc_regularisation_gradients(g, 0.0, 0.0)                      # the configuration's default lambdas: nothing happens
    assert torch.equal(g.opacities.grad, before)


def test_run_mcmc_callbacks_follow_the_schedule():
    """Trainer.py:114-165 under USE_MCMC: SH degree every interval, relocation / growth inside the densification window only, Morton order until its
    end, never an opacity reset."""
    g = _gaussians(200)
    g.active_sh_degree = 0
    sched = dict(D.GARDEN_SCHEDULE, densification_start=2, densification_end=6, densification_interval=2, morton_interval=4, morton_end=8, sh_interval=3,
                 opacity_reset_interval=2)
    gen = torch.Generator().manual_seed(3)
    logits = g.opacities.detach().clone()
    counts, degrees = [], []
    for it in range(10):
        out = D.run_mcmc_callbacks(g, it, sched, cap_max=230, generator=gen, ops=_sim_ops())
        counts.append(None if out is None else out['total'])
        degrees.append(g.active_sh_degree)
    assert [c is not None for c in counts] == [False, False, True, False, True, False, True, False, False, False]      # iterations 2, 4, 6
    assert counts[2] == 210 and counts[4] == 220 and counts[6] == 230                                                   # + 5 % per step up to the cap
    assert degrees[2] == 0 and degrees[3] == 1 and degrees[6] == 2 and degrees[9] == 3
    assert float(g.opacities.detach().max()) >= float(logits.max()) - 1e-6                                             # no reset to logit(0.01)
