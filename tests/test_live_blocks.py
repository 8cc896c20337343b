"""Live-block hand-over (include/fgs_hip.h: fgs_backward_live / fgs_adam_step_multi_live; FasterGSCudaBackend/rasterization.py): the backward
pass flags the blocks of 64 Gaussians without a visible one, the optimizer does not read their (zero) gradients back. Results must be
bit-identical to the plain path, the flags must only ever be used for the very tensors they describe, and every other situation must fall
back. CPU: the simulation build of the same sources; GPU: hardware."""
import numpy as np
import pytest
import torch

import helpers
from harness.scenes import make_s0

ORDER = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')     # optimizer-group order (Model.py:238-245)
LRS = [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3]


def _scene(dev, n=1000, small_image=False):
    """S0 with two contiguous runs of Gaussians behind the camera: several blocks of 64 are entirely invisible, others partly."""
    from harness.scenes import View
    p, v = make_s0(seed=21, n=n)
    p['means'][130:n * 45 // 100, 2] = -30.0
    p['means'][n * 7 // 10:n * 7 // 10 + 33, 2] = -30.0
    if small_image:                                     # the simulation is per-pixel work: the autograd path does not need 128 x 128
        v = View(v.w2c, v.position, 64, 48, 64.0, 64.0, 32.0, 24.0, v.near_plane, v.far_plane, v.background_color)
    return {k: t.to(dev).contiguous() for k, t in p.items()}, v


def _backward(be, dp, RS, dev, live: bool):
    res = be.forward(*[dp[k] for k in helpers.NAMES], RS)
    gi = (torch.randn(3, RS.height, RS.width, generator=torch.Generator().manual_seed(3)) / (3 * RS.height * RS.width)).to(dev)
    n = dp['means'].shape[0]
    flags = torch.full(((n + 63) // 64,), 7, dtype=torch.uint8, device=dev) if live else None
    grads = be.backward(None, gi, res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'],
                        res.buffers, RS, res.state, live_blocks=flags)
    return res, dict(zip(helpers.NAMES, grads)), flags


def _check_kernels(be, dev):
    dp, view = _scene(dev)
    _, RS = helpers.settings_pair(view, device=dev)
    n = dp['means'].shape[0]
    res, g_plain, _ = _backward(be, dp, RS, dev, False)
    res, g_live, flags = _backward(be, dp, RS, dev, True)
    dec = helpers.decode_forward(be, res, n, view.width, view.height)
    vis = np.concatenate([dec['n_touched'] > 0, np.zeros((-n) % 64, bool)]).reshape(-1, 64).any(axis=1)
    assert np.array_equal(flags.cpu().numpy(), vis.astype(np.uint8)) and 0 < vis.sum() < vis.size          # exactly "any visible", both kinds present
    for k in helpers.NAMES:                                      # the gradients themselves: unchanged, dense (two passes: on hardware the float
        if dev == 'cpu':                                         # atomics of the blend-backward kernel add in another order every time)
            assert torch.equal(g_live[k], g_plain[k]), k
        else:
            assert helpers.rel_inf(g_live[k].cpu().numpy(), g_plain[k].cpu().numpy()) < 1e-5, k
    # gradient tensors that are NOT 16-byte aligned (views one float into a buffer): the coalesced 16-byte stores of the gradient kernel must
    # give way to the scalar path, same values
    shapes = [tuple(g_plain[k].shape) for k in helpers.NAMES]
    odd = [torch.empty(int(np.prod(sh)) + 1, dtype=torch.float32, device=dev)[1:].view(sh) for sh in shapes]
    assert all(t.data_ptr() % 16 == 4 for t in odd if t.numel())
    gi = (torch.randn(3, RS.height, RS.width, generator=torch.Generator().manual_seed(3)) / (3 * RS.height * RS.width)).to(dev)
    g_odd = be.backward(None, gi, res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'],
                        res.buffers, RS, res.state, out=tuple(odd))
    for k, t in zip(helpers.NAMES, g_odd):
        if dev == 'cpu':
            assert torch.equal(t, g_live[k]), k
        else:
            assert helpers.rel_inf(t.cpu().numpy(), g_live[k].cpu().numpy()) < 1e-5, k
    dead_rows = torch.from_numpy(np.repeat(~vis, 64)[:n]).to(dev)
    assert dead_rows.any() and all(bool((g_live[k][dead_rows] == 0).all()) for k in helpers.NAMES)

    def adam(grads, live):
        gen = torch.Generator().manual_seed(5)
        P = {k: dp[k].clone() for k in ORDER}
        M = {k: (torch.randn(dp[k].shape, generator=gen) * 1e-3).to(dev) for k in ORDER}
        V = {k: (torch.rand(dp[k].shape, generator=gen) * 1e-6 + 1e-7).to(dev) for k in ORDER}
        for step in (1, 2):
            be.adam_step_multi([grads[k] for k in ORDER], [P[k] for k in ORDER], [M[k] for k in ORDER], [V[k] for k in ORDER], [step] * 6, LRS,
                               0.9, 0.999, 1e-15, live_blocks=live)
        return P, M, V
    ref = adam(g_live, None)                                      # the SAME gradient tensors with and without the promise: the optimizer kernel has no
    got = adam(g_live, flags)                                     # atomics, so this is bit-exact on hardware as well
    # rows of dead blocks are not READ -- except one sentinel float per dead block and tensor (the block's first element): garbage everywhere
    # else in them changes nothing
    first_rows = torch.from_numpy((np.arange(n) % 64 == 0)).to(dev)
    poisoned = {k: g_live[k].clone() for k in ORDER}
    for k in ORDER:
        flat = poisoned[k].reshape(n, -1)
        flat[dead_rows & ~first_rows] = float('nan')
        flat[dead_rows & first_rows, 1:] = float('nan')            # everything but element 0 of the block's first row
    got_poisoned = adam(poisoned, flags)
    for a, b, c in zip(ref, got, got_poisoned):
        for k in ORDER:
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    # a write that no framework bookkeeping sees (p.grad.data.add_(...): hand-written weight decay) makes the sentinels non-zero: those
    # blocks are read after all, and the result is the one of the plain path on the edited gradients
    edited = {k: g_live[k].clone() for k in ORDER}
    for k in ORDER:
        edited[k].data.add_(0.01 * dp[k])
    for a, b in zip(adam(edited, None), adam(edited, flags)):
        for k in ORDER:
            assert torch.equal(a[k], b[k]), ('edited behind the version counter', k)
    # ragged N (last block partial) and a promise that does not fit the tensors
    with pytest.raises(RuntimeError):
        be.adam_step_multi([g_live[k] for k in ORDER], [dp[k].clone() for k in ORDER], [torch.zeros_like(dp[k]) for k in ORDER],
                           [torch.zeros_like(dp[k]) for k in ORDER], [1] * 6, LRS, 0.9, 0.999, 1e-15, live_blocks=flags[:-1])


def test_sim_live_block_kernels():
    _check_kernels(helpers.sim_backend(), 'cpu')


@pytest.mark.gpu
def test_gpu_live_block_kernels(hip_backend):
    _check_kernels(hip_backend, 'cuda')


def _train(dev, be, steps, handover: bool, tamper=None):
    """The reference's loop (Trainer.py:180-199) through the public operators: render -> loss -> backward -> FusedAdam.step -> zero_grad."""
    import FasterGSCudaBackend as FGS
    from FasterGSCudaBackend import rasterization as R
    FGS.set_live_block_handover(handover)
    dp, view = _scene(dev, n=450, small_image=(dev == 'cpu'))
    _, RS = helpers.settings_pair(view, device=dev)
    P = {k: dp[k].clone().requires_grad_(True) for k in ORDER}
    opt = FGS.FusedAdam([{'params': [P[k]], 'lr': lr, 'name': k} for k, lr in zip(ORDER, LRS)], lr=0.0, eps=1e-15)
    for i, k in enumerate(ORDER):                       # non-zero moments (helpers.seeded_moments): two hardware runs differ in the last bits of the
        m0, v0 = helpers.seeded_moments(dp[k].shape, 31 + i)     # gradients, and from zero moments Adam turns a sign change of a tiny gradient into 2 lr
        opt.state[P[k]] = {'step': 0, 'exp_avg': m0.to(dev), 'exp_avg_sq': v0.to(dev)}
    target = torch.rand(3, view.height, view.width, generator=torch.Generator().manual_seed(9)).to(dev)
    stolen = True
    for _ in range(steps):
        image = FGS.diff_rasterize(P['means'], P['scales'], P['rotations'], P['opacities'], P['sh_coefficients_0'], P['sh_coefficients_rest'],
                                   torch.empty(0, device=dev), RS)
        ((image - target) ** 2).mean().backward()
        if handover:
            stolen &= {P[k].grad.data_ptr() for k in ORDER} == {address for address, _ in R._LIVE['slots'][P['means'].data_ptr()]['views']}
        if tamper is not None:
            tamper(P, FGS, RS, target)
        opt.step()
        opt.zero_grad()
    FGS.set_live_block_handover(True)
    return {k: P[k].detach().clone() for k in ORDER}, stolen


def _check_handover(dev, be, monkeypatch):
    import FasterGSCudaBackend as FGS
    from FasterGSCudaBackend import adam as A, rasterization as R
    if dev == 'cpu':                                   # the public operators refuse CPU tensors (no CPU implementation): point them at the simulation
        monkeypatch.setattr(R, '_require_gpu', lambda t: None)
        monkeypatch.setattr(R, 'default_backend', lambda: be)
        monkeypatch.setattr(A, 'default_backend', lambda: be)
    base = FGS.live_block_stats()
    dense, _ = _train(dev, be, 2, False)
    assert FGS.live_block_stats() == base
    fast, stolen = _train(dev, be, 2, True)
    assert stolen, 'autograd did not adopt the arena views as .grad'
    s1 = FGS.live_block_stats()
    assert s1['matched'] == base['matched'] + 2 and s1['missed'] == base['missed']
    start, _ = _train(dev, be, 0, False)

    def same(a, b, what):
        for k in ORDER:
            if dev == 'cpu':
                assert torch.equal(a[k], b[k]), (what, k)                      # bit-identical training
            else:                                                              # two hardware runs: the step taken agrees to the float bar
                assert helpers.rel_inf((a[k] - start[k]).cpu().numpy(), (b[k] - start[k]).cpu().numpy()) < 1e-4, (what, k)
    same(fast, dense, 'hand-over')

    def scale_in_place(P, *_):                         # e.g. gradient clipping: the rows of dead blocks stay zero, but nothing proves it
        for k in ORDER:
            P[k].grad.mul_(0.5)

    def replace(P, *_):
        P['means'].grad = P['means'].grad.clone()

    def decay_behind_the_counter(P, *_):               # old-style weight decay on .data: the version counter does not move, the flags still MATCH,
        for k in ORDER:                                # and the optimizer kernel's sentinel check has to notice
            P[k].grad.data.add_(1e-3 * P[k].data)

    def second_backward(P, FGS_, RS, target):          # accumulation: .grad += gradients of another registered backward pass
        image = FGS_.diff_rasterize(P['means'], P['scales'], P['rotations'], P['opacities'], P['sh_coefficients_0'], P['sh_coefficients_rest'],
                                    torch.empty(0, device=dev), RS)
        ((image - target) ** 2).mean().backward()

    for tamper in (scale_in_place, replace, second_backward):
        before = FGS.live_block_stats()
        a, _ = _train(dev, be, 1, True, tamper)
        after = FGS.live_block_stats()
        assert after['matched'] == before['matched'] and after['missed'] == before['missed'] + 1, tamper.__name__
        b, _ = _train(dev, be, 1, False, tamper)
        same(a, b, tamper.__name__)
    before = FGS.live_block_stats()
    a, _ = _train(dev, be, 1, True, decay_behind_the_counter)
    assert FGS.live_block_stats()['matched'] == before['matched'] + 1          # the registry cannot see it ...
    b, _ = _train(dev, be, 1, False, decay_behind_the_counter)
    same(a, b, 'decay_behind_the_counter')                                       # ... the kernel does


def _check_two_models(dev, be, monkeypatch):
    """Two models in one process, their iterations interleaved (backward A, backward B, step A, step B): each optimizer finds the registration of ITS
    model's pass (the registry is keyed by the model's `means`), and both train exactly as they do alone (round-5 verdict: the single-slot registry
    made the first model miss)."""
    import FasterGSCudaBackend as FGS
    from FasterGSCudaBackend import adam as A, rasterization as R
    if dev == 'cpu':
        monkeypatch.setattr(R, '_require_gpu', lambda t: None)
        monkeypatch.setattr(R, 'default_backend', lambda: be)
        monkeypatch.setattr(A, 'default_backend', lambda: be)
    dp, view = _scene(dev, n=450, small_image=(dev == 'cpu'))
    _, RS = helpers.settings_pair(view, device=dev)
    target = torch.rand(3, view.height, view.width, generator=torch.Generator().manual_seed(9)).to(dev)

    def model(shift):
        P = {k: (dp[k] + (shift if k == 'means' else 0.0)).clone().requires_grad_(True) for k in ORDER}
        opt = FGS.FusedAdam([{'params': [P[k]], 'lr': lr, 'name': k} for k, lr in zip(ORDER, LRS)], lr=0.0, eps=1e-15)
        for i, k in enumerate(ORDER):
            m0, v0 = helpers.seeded_moments(dp[k].shape, 31 + i)
            opt.state[P[k]] = {'step': 0, 'exp_avg': m0.to(dev), 'exp_avg_sq': v0.to(dev)}
        return P, opt

    def backward(P):
        image = FGS.diff_rasterize(P['means'], P['scales'], P['rotations'], P['opacities'], P['sh_coefficients_0'], P['sh_coefficients_rest'],
                                   torch.empty(0, device=dev), RS)
        ((image - target) ** 2).mean().backward()

    results = {}
    for interleaved in (False, True):
        FGS.set_live_block_handover(True)
        (Pa, oa), (Pb, ob) = model(0.0), model(0.01)
        before = FGS.live_block_stats()
        for _ in range(2):
            if interleaved:
                backward(Pa); backward(Pb); oa.step(); ob.step()
            else:
                backward(Pa); oa.step(); backward(Pb); ob.step()
            oa.zero_grad(); ob.zero_grad()
        after = FGS.live_block_stats()
        assert after['matched'] == before['matched'] + 4 and after['missed'] == before['missed'], (interleaved, before, after)
        assert not R._LIVE['slots']                                           # every registration was consumed by its owner
        results[interleaved] = ({k: Pa[k].detach().clone() for k in ORDER}, {k: Pb[k].detach().clone() for k in ORDER})
    if dev == 'cpu':
        for m in (0, 1):
            for k in ORDER:
                assert torch.equal(results[True][m][k], results[False][m][k]), (m, k)


def test_sim_two_models_keep_their_own_registration(monkeypatch):
    _check_two_models('cpu', helpers.sim_backend(), monkeypatch)


@pytest.mark.gpu
def test_gpu_two_models_keep_their_own_registration(hip_backend, monkeypatch):
    _check_two_models('cuda', hip_backend, monkeypatch)


def test_async_forward_scopes_keep_their_own_tables():
    """`async_forward_scope`: switch, per-view ratios and the overflow mark of a scope are invisible to the others (two models over the same views)."""
    import FasterGSCudaBackend as FGS
    from FasterGSCudaBackend import rasterization as R
    assert FGS.async_forward_stats()['enabled'] is False
    with FGS.async_forward_scope('model_b'):
        FGS.set_async_forward(True, headroom=1.5)
        R._ASYNC['per_view']['some view'] = (2.0, lambda: None, 0)
        R._ASYNC['step_invalid'] = frozenset({1234})
        assert FGS.async_forward_stats()['enabled'] and FGS.async_forward_stats()['views'] == 1
    assert FGS.async_forward_stats()['enabled'] is False and FGS.async_forward_stats()['views'] == 0 and not FGS.take_async_overflow()
    with FGS.async_forward_scope('model_b'):
        assert FGS.async_forward_stats()['headroom'] == 1.5 and FGS.async_forward_stats()['views'] == 1       # scopes persist by name
        assert FGS.take_async_overflow() and not FGS.take_async_overflow()
    FGS.async_forward_scope.drop('model_b')
    with FGS.async_forward_scope('model_b'):
        assert FGS.async_forward_stats()['enabled'] is False


def test_sim_handover_through_autograd(monkeypatch):
    _check_handover('cpu', helpers.sim_backend(), monkeypatch)


@pytest.mark.gpu
def test_gpu_handover_through_autograd(hip_backend, monkeypatch):
    _check_handover('cuda', hip_backend, monkeypatch)
