"""GPU tests (-m gpu) of BASELINE.json configs[3]: fgs_backward_adam_fused against the ORACLE's backward -> adam_step
(reference contract: torch_bindings/adam.py:11-36, adam/src/adam.cu:22-33, kernels_backward.cuh:15-257).

Every comparison is flip-aware (helpers.check_flip_aware): the oracle names the pixels / Gaussians that sit within 5e-6 of the
alpha >= 1/255 test (or on a preprocess floor / ceil boundary); their number is bounded (< 1e-3 of all) and everything else --
parameters (as the step they took), exp_avg, exp_avg_sq of all six groups and densification_info -- must agree to 1e-4.
Moments start non-zero so that (a) invisible Gaussians show the momentum-only update and decay (adam.py:16: dense zero
gradients) and (b) the first Adam step is well conditioned (with m = v = 0 it is lr * sign(g), discontinuous at g = 0)."""
import numpy as np
import pytest
import torch

import helpers
from harness.scenes import View, make_garden_like, make_s0, orbit_views

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ORDER = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')      # Model.py:238-245
GRAD_OF = {'means': 'means', 'sh_coefficients_0': 'sh0', 'sh_coefficients_rest': 'sh_rest', 'opacities': 'opacities',
           'scales': 'scales', 'rotations': 'rotations'}
LRS = [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3]


def _odd(t):
    """The same values in a tensor that starts 4 bytes past a 16-byte boundary (a view one float into a buffer)."""
    o = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)[1:].view(t.shape)
    o.copy_(t)
    assert t.numel() == 0 or o.data_ptr() % 16 == 4
    return o


def _run(hip_backend, oracle, params, view, K=16, aa=False, steps=3, tol=1e-4, label='', single_kernel=True, masked_budget=None, unaligned=False, near_tol=None, skip_on_new_ties=False):
    S, RS = helpers.settings_pair(view, K, aa, device=DEV)
    n = params['means'].shape[0]
    gen = torch.Generator().manual_seed(7)
    P0 = {k: params[k].clone() for k in ORDER}
    M0 = {k: torch.randn(params[k].shape, generator=gen) * 1e-3 for k in ORDER}
    V0 = {k: torch.rand(params[k].shape, generator=gen) * 1e-6 for k in ORDER}
    dP, dM, dV = ({k: d[k].to(DEV).contiguous().clone() for k in ORDER} for d in (P0, M0, V0))
    if unaligned:
        dP, dM, dV = ({k: _odd(d[k]) for k in ORDER} for d in (dP, dM, dV))
    oP, oM, oV = ({k: np.ascontiguousarray(d[k].numpy().copy()) for k in ORDER} for d in (P0, M0, V0))
    gi = torch.randn(3, view.height, view.width, generator=gen) / (view.height * view.width)
    gi_np, gi_dev = gi.numpy(), gi.to(DEV)
    dens_dev, dens_o = torch.zeros(2, n, device=DEV), np.zeros((2, n), np.float32)
    masked = np.zeros(n, bool)
    near = np.zeros(n, bool)
    new_ties = 0          # first step at which device and oracle ordered Gaussians with EQUAL depth keys differently (skip_on_new_ties)
    ever_visible = np.zeros(n, bool)
    if not single_kernel:                      # the round-1 two-kernel form exists in the dev library only (fgs_debug_set_option)
        assert hip_backend.lib.fgs_debug_set_option(3, 0) == 0
    try:
        for step in range(1, steps + 1):
            res = hip_backend.forward(*[dP[k] for k in helpers.NAMES], RS)
            dec = helpers.decode_forward(hip_backend, res, n, view.width, view.height) if n <= 200_000 else None
            hip_backend.backward_adam_fused(dens_dev, gi_dev, res.image, [dP[k] for k in ORDER], [dM[k] for k in ORDER],
                                            [dV[k] for k in ORDER], res.buffers, RS, res.state, step, LRS)
            f = oracle.forward(*[oP[k] for k in helpers.NAMES], S, bucket_size=64)
            if skip_on_new_ties and step > 1 and dec is not None and dec['V'] == f['V'] and dec['I'] == f['I'] and not np.array_equal(dec['inst_prims'], f['inst_prims']):
                # The scenes start without equal depth keys (helpers.fuzz_configuration nudges them apart), but an Adam step can MAKE a tie (2 000 random
                # depths in one float32 binade collide readily), and tied Gaussians keep K1's atomic arrival order on the device, the index order in the
                # oracle (kf:204-208 has the same freedom): from there on the two runs blend two overlapping Gaussians in different orders. Seed 6447
                # of a 4000-seed sweep (round 6): tools/diag_fused_fuzz_seed.py. Verified to be exactly that -- same sets per tile, same keys -- and skipped.
                try:
                    helpers.check_lists_up_to_ties(dec, f)
                    tied = True
                except AssertionError:
                    tied = False
                if tied and not new_ties:
                    new_ties = step
            masks = helpers.flip_masks(oracle, f, S, dec)
            masked |= masks['prim']
            near |= masks['near']
            ever_visible |= f['n_touched'] > 0
            g = oracle.backward(f, S, gi_np, dens_o)
            for k, lr in zip(ORDER, LRS):
                oracle.adam_step(np.ascontiguousarray(g[GRAD_OF[k]].reshape(oP[k].shape)), oP[k], oM[k], oV[k], step, lr)
            del res
    finally:
        if not single_kernel:
            hip_backend.lib.fgs_debug_set_option(3, 1)
    if DEV != 'cpu':
        torch.cuda.synchronize()
    def compare():
        nonlocal near
        assert masked.mean() < (1e-3 * steps + 2.0 / n if masked_budget is None else masked_budget), (label, 'masked Gaussians', float(masked.mean()))
        keep = ~masked
        if near_tol is not None:
            # the adversarial fuzz scenes only (as helpers.check_flip_aware): Gaussians that merely blend into a pixel with a borderline pair see their share
            # of that pixel scaled by 1 - 1/255 when the pair flips; they are held to `near_tol` instead (seed 6447 of a 4000-seed sweep, round 6)
            near &= keep
            keep = keep & ~near
            assert near.mean() < 0.25, (label, 'share of Gaussians in the near class', float(near.mean()))
        report = {}
        for k in ORDER:
            if oP[k].size == 0:
                continue
            start = P0[k].numpy()
            moved_ref, moved = oP[k] - start, dP[k].cpu().numpy() - start
            assert np.abs(moved_ref).max() > 0
            report[k] = (helpers.masked_rel_inf(moved, moved_ref, keep), helpers.masked_rel_inf(dM[k].cpu().numpy(), oM[k], keep),
                         helpers.masked_rel_inf(dV[k].cpu().numpy(), oV[k], keep))
            assert max(report[k]) < tol, (label, k, report)
            if near_tol is not None and near.any():
                report[k + '_near'] = (helpers.masked_rel_inf(moved, moved_ref, near), helpers.masked_rel_inf(dM[k].cpu().numpy(), oM[k], near),
                                       helpers.masked_rel_inf(dV[k].cpu().numpy(), oV[k], near))
                assert max(report[k + '_near']) < near_tol, (label, k + ' (Gaussians behind a borderline pair)', report)
            # element by element (helpers.elementwise_fraction): the step each parameter took and both moments
            elem = (helpers.elementwise_fraction(moved, moved_ref, keep, kind='elementwise_step_' + k),
                    helpers.elementwise_fraction(dM[k].cpu().numpy(), oM[k], keep, kind='elementwise_exp_avg_' + k),
                    helpers.elementwise_fraction(dV[k].cpu().numpy(), oV[k], keep, kind='elementwise_exp_avg_sq_' + k))
            report[k + '_elem'] = elem
            assert max(elem) < max(helpers.ELEM_FRACTION, 2.0 * oP[k][0].size / oP[k].size), (label, k, 'element-wise 1e-4', report)
            assert helpers.rel_inf(dM[k].cpu().numpy(), oM[k]) < 5e-2, (label, k, 'masked')
        report['dens'] = helpers.masked_rel_inf(dens_dev.cpu().numpy().T, dens_o.T, keep)
        assert report['dens'] < tol, (label, report)
        # invisible Gaussians: zero gradient, yet the moments decay and the parameters move by momentum (adam.py:16)
        inv = ~ever_visible & keep          # unseen in EVERY step (the parameters move: seed 572 of a wide sweep has a Gaussian that leaves through the far
                                            # plane after step 1), and not on a cull threshold (visible to one side only)
        if inv.any():
            k = 'means'
            assert np.abs(dM[k].cpu().numpy()[inv] - M0[k].numpy()[inv] * 0.9 ** steps).max() < 1e-5 * np.abs(M0[k].numpy()).max()   # fp32: m * 0.9 * 0.9 ...
            assert np.abs(dP[k].cpu().numpy()[inv] - P0[k].numpy()[inv]).max() > 0
        return report

    try:
        return compare()
    except AssertionError:
        if new_ties:      # the comparison is not meaningful from that step on; without a failure the tie was harmless (the tied Gaussians do not overlap)
            pytest.skip(f'{label}: step {new_ties}: an Adam step produced equal depth keys; device and oracle blend the tied Gaussians in different orders')
        raise


def test_fused_s0_three_steps(hip_backend, oracle):
    """configs[0] scene; 1000 Gaussians = 15 full waves + a ragged one; 100 of them behind the camera."""
    params, view = make_s0()
    params['means'][:100, 2] = -10.0
    _run(hip_backend, oracle, params, view, label='S0')


@pytest.mark.parametrize('n,w,h,K,aa,sh_bases', [(777, 50, 30, 16, False, 16), (301, 48, 36, 9, True, 16), (130, 130, 25, 4, False, 16),
                                                  (63, 16, 12, 1, False, 16), (257, 64, 48, 4, True, 4), (66, 40, 30, 1, False, 1)])
def test_fused_ragged_sizes_sh_degrees_antialiasing(hip_backend, oracle, n, w, h, K, aa, sh_bases):
    """N not divisible by 4 / 64, SH degrees 0-3 (active) on full-size and on smaller sh_coefficients_rest tensors, partial tiles."""
    p, v = make_s0(seed=3, n=n, sh_bases=sh_bases)
    p['means'][: n // 10, 2] = -10.0
    v = View(v.w2c, v.position, w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, 0.2, 1e4, torch.tensor([0.2, 0.5, 0.7]))
    _run(hip_backend, oracle, p, v, K, aa, steps=2, label=f'n{n}')


def test_fused_unaligned_parameters_and_moments(hip_backend, oracle):
    """Parameters and both moments of all six groups start 4 bytes past a 16-byte boundary: the kernel's 16-byte accesses (phase A's staged
    blocks, phase B's SH-rest pieces) have to give way to the scalar path -- same values."""
    p, v = make_s0(seed=5, n=333)
    p['means'][:30, 2] = -10.0
    _run(hip_backend, oracle, p, v, steps=2, label='unaligned', unaligned=True)


def test_fused_two_kernel_form_s0(hip_dev_backend, oracle):
    """The round-1 two-kernel form (fgs_debug_set_option(3, 0), libfgs_hip_dev.so) stays selectable for A/B and must meet the same bar."""
    hip_backend = hip_dev_backend
    params, view = make_s0()
    params['means'][:100, 2] = -10.0
    _run(hip_backend, oracle, params, view, steps=2, label='S0 two-kernel', single_kernel=False)


def test_fused_mid_size(hip_backend, oracle):
    """60 k garden-like Gaussians at 640x360 (the mid-size scene of test_gpu_parity.py), three steps."""
    params = make_garden_like(60_001)
    params['scales'] = params['scales'] + 0.7
    v = orbit_views(8, width=640, height=360, focal=473.0)[1]
    _run(hip_backend, oracle, params, v, label='60k')


def test_fused_full_size_1m_1080p(hip_backend, oracle):
    """1 000 003 Gaussians at 1920x1080 (ragged N; 45 M SH-rest floats per stream), two steps against the oracle."""
    params = make_garden_like(1_000_003)
    v = orbit_views(8)[2]
    _run(hip_backend, oracle, params, v, steps=2, label='1M')


def test_fused_full_size_s2_3m_1080p(hip_backend, oracle):
    """BASELINE.json configs[3] at the size bench.py times it (`fused_train_iters_per_sec`: S2, 3 M Gaussians, 1920x1080; VERDICT r2
    missing #2): two fused backward+Adam steps against oracle backward -> oracle Adam on all 59 x 3 M floats and both moments."""
    params = make_garden_like(3_000_000)
    _run(hip_backend, oracle, params, orbit_views(8)[0], steps=2, label='S2 fused')
