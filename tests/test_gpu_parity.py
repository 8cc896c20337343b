"""GPU tests (-m gpu): the HIP library, called through the C ABI and the public operators, against the CPU oracle.

Tolerances (BASELINE.json north_star): integer / index work bit-exact; floats within 1e-4 relative -- stated per test.
Two facts bound what "bit-exact" can mean on real hardware ('fast-math parity'):
 * preprocess is built without FMA contraction and with IEEE div/sqrt, so it differs from the oracle only through
   ULP-level differences of expf/logf (ocml vs glibc); a screen bound / exact tile count can flip for the rare primitive
   whose floor/ceil/threshold input lies within an ULP of an integer. On the fixed scenes of this file that never happens
   (round 3: asserted, _forward_check); the large and the randomised scenes (flip-aware tests) tolerate and mask such primitives.
 * the blend kernels use the hardware exp (v_exp_f32) and FMA contraction; an alpha within ~1e-6 relative of the 1/255
   threshold can be kept on one side and dropped on the other, changing that pixel by up to ~4e-3. The oracle NAMES the pixels
   (and Gaussians) that own such a pair (oracle.threshold_risk -> helpers.flip_masks); per-pixel outputs (image, final_T,
   n_processed) and gradients are held to 1e-4 / exact OUTSIDE that mask, the mask is bounded to 1e-3 of all entries and masked
   pixels to 5e-3. No test compares against a count of unexplained outliers; achieved errors per assert site on an MI355X are in
   profiles/archive/r02_gpu_tolerance_slack.txt (FGS_TOL_LOG=<file> regenerates it).
"""
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers
from harness.scenes import View, make_garden_like, make_s0, orbit_views

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / 'golden'
DEV = 'cuda'


def _to(params, dev=None):
    return {k: v.to(dev or DEV).contiguous() for k, v in params.items()}


def _grads_close(grads, g, tol=1e-4, truth=None, free_rows=2):
    """Max-norm 1e-4 per tensor; with `truth` (oracle.forward_backward_f64 of the same scene) also the element-wise 1e-4 bar, three-way
    (helpers.elementwise_three_way). These scenes are compared without the oracle's threshold-risk mask, so up to `free_rows` Gaussians
    (one alpha-test flip) may exceed the element-wise bar -- on the small scenes only: from 60 k Gaussians on nothing is free (round 4)."""
    if g[helpers.GRAD_KEYS[0]].shape[0] >= helpers.LARGE_SCENE_ROWS:
        free_rows = 0
    for k, t in zip(helpers.GRAD_KEYS, grads):
        a = t.detach().cpu().numpy().reshape(g[k].shape)
        e = helpers.rel_inf(a, g[k])
        assert e < tol, (k, e)
        if truth is not None:
            fh, fo, cnt = helpers.elementwise_three_way(a, g[k], np.asarray(truth[k]).reshape(g[k].shape), kind=k, free_rows=free_rows)
            assert helpers.three_way_ok(fh, fo, cnt, cluster=int(np.prod(g[k].shape[1:])) if g[k].ndim > 1 else 1), (k, 'element-wise 1e-4 (HIP vs fp64, oracle32 vs fp64, entries)', fh, fo, cnt)


def test_wave_primitives_selftest(hip_backend):
    """DPP wave_shr:1 / wave_rol:1 direction, ballot prefix, readlane, reductions on the real wave64."""
    out = torch.zeros(256, dtype=torch.int32, device=DEV)
    assert hip_backend.lib.fgs_debug_wave_selftest(out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    o, l = out.cpu().numpy(), np.arange(64)
    assert np.array_equal(o[:64], np.where(l == 0, 1000, 99 + l)), o[:64]
    assert np.array_equal(o[64:128], 1000 + (l + 1) % 64), o[64:128]
    assert np.array_equal(o[128:192], (l + 2) // 3) and np.all(o[192:] == 2142), (o[128:192], o[192:196])


def _forward_check(hip_backend, oracle, params, view, K=16, aa=False, bg=None, int_budget=False, max_masked_pixels=1e-3):
    """Every forward intermediate against the oracle. The integer ones (screen bounds, tile counts and everything derived from them) are
    compared bit for bit: on the fixed scenes of this file no primitive's bounds differ from the oracle's on an MI355X (logged per call under
    FGS_TOL_LOG, `profiles/archive/r03_gpu_tolerance_slack.txt` part 3: 16 of 16), so the libm-ULP budget of rounds 1-2 -- up to max(1, n / 1000)
    primitives, after which the downstream exact comparisons were dropped -- is now opt-in (`int_budget`) and used by no fixed scene."""
    S, RS = helpers.settings_pair(view, K, aa, bg, device=DEV)
    dp = _to(params)
    n = dp['means'].shape[0]
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    torch.cuda.synchronize()
    f = oracle.forward(*helpers.np_params(params), S, bucket_size=64)
    dec = helpers.decode_forward(hip_backend, res, n, view.width, view.height)
    vis = f['n_touched'] > 0
    bad = int((dec['n_touched'] != f['n_touched']).sum()) + int((dec['screen_bounds'][vis] != f['screen_bounds'][vis]).any(axis=1).sum())
    helpers.log_note('int_mismatch_primitives', bad, n=n)
    assert bad == 0 or int_budget, ('integer intermediates differ from the oracle', bad, n)
    budget = 0 if bad == 0 else max(1, n // 1000)
    pixel_mask = helpers.flip_masks(oracle, f, S, dec)['pixel']      # pixels within an ULP-scale margin of the alpha / T thresholds
    if bad == 0:
        helpers.check_forward_against_oracle(dec, f, False, view.width, view.height, res.image.cpu().numpy(), pixel_mask=pixel_mask,
                                             max_masked_pixels=max_masked_pixels)
    else:   # an ULP-level flip in a bound: V/I may differ by a few; the image is still held to 1e-4 outside the risk mask, which
        assert bad <= budget, bad                                    # then also has to cover the pixels of the differing Gaussians
        assert abs(dec['I'] - f['I']) <= 8 * budget
        img = res.image.cpu().numpy()
        err = np.abs(img.astype(np.float64) - f['image']).max(axis=0)
        outside = err[~pixel_mask] > 1e-4 * max(1.0, float(np.abs(f['image']).max()))
        assert float(pixel_mask.mean()) < 1e-3 and float(outside.mean()) < 1e-5 * bad + 1e-6, (float(pixel_mask.mean()), float(outside.mean()), bad)
    return res, f, dp, RS, S


def test_s0_every_intermediate(hip_backend, oracle):
    """configs[0] scene: preprocess / sort / instance / range / bucket intermediates bit-exact, floats 1e-5, image 1e-4."""
    params, view = make_s0()
    res, f, *_ = _forward_check(hip_backend, oracle, params, view)
    assert res.state[0] == f['V']


def test_s0_against_committed_golden(hip_backend):
    """No oracle run: the committed fixture (tests/golden/s0.npz) pins image, integer intermediates and gradients."""
    g = np.load(GOLDEN / 's0.npz')
    params = {k: torch.from_numpy(g[f'in_{k}']) for k in helpers.NAMES}
    _, view = make_s0()
    _, RS = helpers.settings_pair(view, device=DEV)
    dp = _to(params)
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(hip_backend, res, 1000, 128, 128)
    assert dec['V'] == int(g['V']) and dec['I'] == int(g['I'])
    assert np.array_equal(dec['n_touched'], g['n_touched']) and np.array_equal(dec['inst_keys'], g['inst_keys'])
    assert np.array_equal(dec['inst_prims'], g['inst_prims']) and np.array_equal(dec['ranges'], g['ranges'])
    assert np.array_equal(dec['bucket_offsets'], g['b64_bucket_offsets'])
    assert helpers.rel_inf(res.image.cpu().numpy(), g['image']) < 1e-4
    dens = torch.zeros(2, 1000, device=DEV)
    grads = hip_backend.backward(dens, torch.from_numpy(g['grad_image']).to(DEV), res.image, dp['means'], dp['scales'], dp['rotations'],
                                 dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, {k: g[f'grad_{k}'] for k in helpers.GRAD_KEYS})
    assert helpers.rel_inf(dens.cpu().numpy(), g['densification_info']) < 1e-4


@pytest.mark.parametrize('w,h,K,aa', [(48, 36, 4, True), (50, 30, 16, False), (16, 12, 1, False), (130, 25, 9, True), (333, 211, 16, False)])
def test_partial_tiles_sh_degrees_antialiasing(hip_backend, oracle, w, h, K, aa):
    p, v = make_s0(seed=3, n=300)
    v = View(v.w2c, v.position, w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, 0.2, 1e4, torch.tensor([0.2, 0.5, 0.7]))
    res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v, K, aa)
    gi = np.random.default_rng(5).standard_normal(f['image'].shape).astype(np.float32)
    dens_o = np.zeros((2, 300), np.float32)
    g = oracle.backward(f, S, gi, dens_o)
    dens = torch.zeros(2, 300, device=DEV)
    grads = hip_backend.backward(dens, torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'],
                                 dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, g, truth=oracle.forward_backward_f64(f, S, gi))
    assert helpers.rel_inf(dens.cpu().numpy(), dens_o) < 1e-4


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4, 5])
def test_blend_backward_variants_agree_with_oracle(hip_dev_backend, oracle, variant):
    """All formulations of K11 (0 / 2 systolic lane = Gaussian, 1 strip lane = pixel, 3 the product's, 4 lane = pixel + matrix cores, 5 the product's with the items of a wave chained through the lanes) against the oracle
    on a deep scene -- on libfgs_hip_dev.so: the product library carries variant 3 only."""
    hip_backend = hip_dev_backend
    p, v = make_s0(seed=11, n=1500)
    p['means'][:, :2] *= 0.3
    hip_backend.lib.fgs_debug_set_backward_variant(variant)
    try:
        res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v)
        gi = np.random.default_rng(9).standard_normal(f['image'].shape).astype(np.float32)
        g = oracle.backward(f, S, gi)
        grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                     dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
        _grads_close(grads, g, truth=oracle.forward_backward_f64(f, S, gi))
    finally:
        hip_backend.lib.fgs_debug_set_backward_variant(3)


def test_equal_depth_keys_keep_every_order_independent_quantity(hip_backend, oracle):
    """Hundreds of exactly equal depth keys (the fuzz scenes avoid ties): on hardware their blending order is K1's atomic compaction order, as in
    the reference (kf:204-208) -- image and gradients legitimately depend on it, everything else must agree with the oracle exactly
    (helpers.check_order_independent_quantities), and the image stays within what reordering equal-depth layers can do."""
    p, view = helpers.tied_depth_scene()
    S, RS = helpers.settings_pair(view, device=DEV)
    dp = _to(p)
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    torch.cuda.synchronize()
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    dec = helpers.decode_forward(hip_backend, res, p['means'].shape[0], view.width, view.height)
    helpers.check_order_independent_quantities(dec, f, view.width, view.height)
    assert float(np.abs(res.image.cpu().numpy() - f['image']).max()) < 0.5          # a different order of tied layers, not a different scene


def test_equal_depth_keys_image_and_gradients_in_the_device_order(hip_backend, oracle):
    """The other half of the tied-depth contract (round-5 verdict: no test held image / gradient parity on a scene with ties). Among equal depth keys
    the blending order is whatever the visible list's order was (K1's atomic compaction on hardware, kf:204-208 in the reference; the index order in the
    oracle). So: read the order the DEVICE chose out of its buffers, hand the oracle the same Gaussians re-indexed in that order -- its stable sort
    then reproduces the device's order exactly -- and hold image, final transmittance, per-pixel contributor counts and all six gradients to the usual
    bars. Nothing about this scene is order-independent any more; the comparison is."""
    p, view = helpers.tied_depth_scene()
    S, RS = helpers.settings_pair(view, device=DEV)
    dp = _to(p)
    n = p['means'].shape[0]
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    gi = np.random.default_rng(4).standard_normal((3, view.height, view.width)).astype(np.float32)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'], dp['rotations'],
                                 dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    torch.cuda.synchronize()
    dec = helpers.decode_forward(hip_backend, res, n, view.width, view.height)
    f0 = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    sel = 0 if np.array_equal(np.sort(dec['prim_idx0']), np.sort(f0['prim_idx'])) and np.all(np.diff(dec['depth_keys0'].astype(np.int64)) >= 0) else 1
    device_order = dec[f'prim_idx{sel}'].astype(np.int64)                       # the visible Gaussians, front to back, ties as the device broke them
    assert len(np.unique(f0['depth_keys'])) < f0['V'] // 50 and not np.array_equal(device_order, f0['prim_idx'])      # heavily tied, and broken differently
    perm = np.concatenate([device_order, np.setdiff1d(np.arange(n), device_order)])
    pp = {k: v[torch.from_numpy(perm)].contiguous() for k, v in p.items()}
    f = oracle.forward(*helpers.np_params(pp), S, bucket_size=64)
    assert np.array_equal(perm[f['prim_idx']], device_order)                   # the oracle now walks the device's order
    assert helpers.rel_inf(res.image.cpu().numpy(), f['image']) < 1e-4
    assert np.abs(helpers.tiles_to_image(dec['final_T_tiles'], view.width, view.height).reshape(-1) - f['final_T']).max() < 1e-5
    assert np.array_equal(helpers.tiles_to_image(dec['n_processed_tiles'], view.width, view.height).reshape(-1), f['n_processed'])
    g = oracle.backward(f, S, gi)
    inverse = np.empty(n, np.int64); inverse[perm] = np.arange(n)
    g = {k: g[k][inverse] for k in helpers.GRAD_KEYS}                           # back to the caller's indexing
    _grads_close(grads, g)


def test_uninitialised_scratch_is_harmless(hip_dev_backend, oracle):
    """Same as the simulation test: 0xFF-poisoned scratch must not reach any output (NaN checkpoints of finished pixels); every K11 formulation
    of the dev library (variant 3 = the product's kernel, same source)."""
    hip_backend = hip_dev_backend
    p, v = make_s0(seed=11, n=1500)
    p['means'][:, :2] *= 0.15
    p['opacities'] -= 2.5
    p['means'][:50, 2] = -10.0
    be = helpers.poisoned(hip_backend)
    res, f, dp, RS, S = _forward_check(be, oracle, p, v)
    gi = np.random.default_rng(9).standard_normal(f['image'].shape).astype(np.float32)
    g = oracle.backward(f, S, gi)
    truth = oracle.forward_backward_f64(f, S, gi)
    for variant in (0, 1, 2, 3, 4, 5):
        be.lib.fgs_debug_set_backward_variant(variant)
        try:
            grads = be.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
            assert all(bool(torch.isfinite(t).all()) for t in grads)
            _grads_close(grads, g, truth=truth)
        finally:
            be.lib.fgs_debug_set_backward_variant(3)
    # the fused backward + Adam over poisoned buffers as well: lr = 1 and zero moments turn the first step into p - sign(g), so the parameters that
    # moved the wrong way (or did not stay put) name every gradient sign the fused kernel got wrong
    order = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
    key_of = dict(zip(order, ('means', 'sh0', 'sh_rest', 'opacities', 'scales', 'rotations')))
    res2 = be.forward(*[dp[k] for k in helpers.NAMES], RS)
    P = [dp[k].clone() for k in order]
    M, V = [torch.zeros_like(t) for t in P], [torch.zeros_like(t) for t in P]
    be.backward_adam_fused(None, torch.from_numpy(gi).to(DEV), res2.image, P, M, V, res2.buffers, RS, res2.state, 1, [1.0] * 6)
    for k, t, m in zip(order, P, M):
        assert bool(torch.isfinite(t).all()) and bool(torch.isfinite(m).all()), k
        ref = g[key_of[k]].reshape(tuple(t.shape))
        assert helpers.rel_inf(m.cpu().numpy(), 0.1 * ref) < 1e-4, k                 # exp_avg after one step = (1 - beta1) * gradient


def test_large_footprints_and_long_lists(hip_backend, oracle):
    p, v = make_s0(seed=7, n=200)
    p['scales'] = p['scales'] + 2.3
    _forward_check(hip_backend, oracle, p, v)
    p, v = make_s0(seed=11, n=1500)
    p['means'][:, :2] *= 0.15
    p['opacities'] -= 2.5
    res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v)
    gi = np.ones_like(f['image'])
    g = oracle.backward(f, S, gi)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                 dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, g, truth=oracle.forward_backward_f64(f, S, gi))


def test_huge_footprints_workgroup_path(hip_backend, oracle):
    p, v = make_s0(seed=13, n=40)
    v = View(v.w2c, v.position, 640, 480, 500.0, 500.0, 320.0, 240.0, 0.2, 1e4, torch.zeros(3))
    p['scales'][:6] = p['scales'][:6] + 3.2
    p['scales'][6:20] = p['scales'][6:20] + 1.8
    _forward_check(hip_backend, oracle, p, v)


def test_hot_footprints_accumulate_through_replicas(hip_backend, oracle):
    """Footprints above 256 candidate tiles use K11's replicated accumulators (and > 1024 the workgroup path of K1 / K5): forward
    intermediates and all gradients against the oracle, with more hot Gaussians than one wave holds."""
    p, v = make_s0(seed=13, n=400)
    v = View(v.w2c, v.position, 960, 540, 700.0, 700.0, 480.0, 270.0, 0.2, 1e4, torch.zeros(3))
    p['scales'][:90] = p['scales'][:90] + 2.2
    p['scales'][:12] = p['scales'][:12] + 1.0
    p['opacities'][:90] -= 1.5
    res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v)
    sb = f['screen_bounds'].astype(np.int64)
    n_max = ((sb[:, 1] + 15) // 16 - sb[:, 0] // 16) * ((sb[:, 3] + 11) // 12 - sb[:, 2] // 12)
    assert ((n_max > 256) & (f['n_touched'] > 0)).sum() > 64 and ((n_max > 1024) & (f['n_touched'] > 0)).sum() >= 4
    gi = np.random.default_rng(9).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    g = oracle.backward(f, S, gi)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                 dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, g, truth=oracle.forward_backward_f64(f, S, gi))


def test_known_answer_single_gaussian_on_device(hip_backend):
    """The hand-derived known answer of tests/test_oracle.py::test_known_answer_single_gaussian, against the HIP path directly
    (no oracle involved): one isotropic Gaussian on the optical axis, image and two analytic gradients."""
    W, H, f, z, sigma, logit = 32, 24, 30.0, 5.0, 0.1, 2.0
    bg, sh0, C0 = np.array([0.1, 0.2, 0.3]), np.array([0.7, -0.2, 1.1]), 0.28209479177387814
    v = View(torch.eye(4), torch.zeros(3), W, H, f, f, W / 2, H / 2, 0.2, 1e4, torch.tensor(bg, dtype=torch.float32))
    _, RS = helpers.settings_pair(v, 1, False, device=DEV)
    t = lambda x: torch.tensor(x, dtype=torch.float32, device=DEV)
    P = dict(means=t([[0, 0, z]]), scales=t(np.full((1, 3), np.log(sigma))), rotations=t([[1, 0, 0, 0]]), opacities=t([[logit]]),
             sh_coefficients_0=t(sh0.reshape(1, 1, 3)), sh_coefficients_rest=torch.zeros(1, 15, 3, device=DEV))
    res = hip_backend.forward(*[P[k] for k in helpers.NAMES], RS)
    cov, op = (f * sigma / z) ** 2 + 0.3, 1.0 / (1.0 + np.exp(-logit))
    ys, xs = np.mgrid[0:H, 0:W]
    G = np.exp(-0.5 * ((xs + 0.5 - W / 2) ** 2 + (ys + 0.5 - H / 2) ** 2) / cov)
    alpha = np.where(op * G >= 1.0 / 255.0, op * G, 0.0)
    colour = 0.5 + C0 * sh0
    expected = alpha[None] * colour[:, None, None] + (1.0 - alpha[None]) * bg[:, None, None]
    assert res.state[0] == 1 and np.abs(res.image.cpu().numpy() - expected).max() < 5e-6
    gi = np.random.default_rng(0).standard_normal((3, H, W)).astype(np.float32)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, P['means'], P['scales'], P['rotations'],
                                 P['opacities'], P['sh_coefficients_rest'], res.buffers, RS, res.state)
    d_sh0 = C0 * (alpha[None] * gi).sum(axis=(1, 2))
    d_logit = op * (1.0 - op) * (np.where(alpha > 0, G, 0.0)[None] * (colour - bg)[:, None, None] * gi).sum()
    assert np.abs(grads[4].cpu().numpy().reshape(3) - d_sh0).max() < 1e-4 * max(1.0, np.abs(d_sh0).max())
    assert abs(float(grads[3].cpu().reshape(-1)[0]) - d_logit) < 1e-4 * max(1.0, abs(d_logit))


def test_more_than_65536_tiles_uses_32_bit_keys(hip_backend, oracle):
    """fwd:152-153: above 65 536 tiles the instance keys are 32-bit (275 x 250 = 68 750 tiles here). Forward intermediates and the
    backward pass against the oracle."""
    p, v = make_s0(seed=17, n=4000)
    v = View(v.w2c, v.position, 4400, 3000, 3500.0, 3500.0, 2200.0, 1500.0, 0.2, 1e4, torch.zeros(3))
    res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v)
    assert int(f['inst_keys'].max()) > 65535
    gi = np.random.default_rng(4).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    g = oracle.backward(f, S, gi)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                 dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, g, tol=2e-4, truth=oracle.forward_backward_f64(f, S, gi))


def test_tile_columns_beyond_1024_use_escape_rows(hip_backend, oracle):
    """Footprint rows (fgs_math.h) hold the tile box origin in 10 bits per axis: Gaussians whose boxes start at tile column >= 1024 (image wider than
    16 384 px) travel as ESCAPE rows and are re-tested by the instance kernel from the record. 20 000 x 36 px = 1 250 x 3 tiles, Gaussians spread over the
    whole width: every forward intermediate bit-exact against the oracle, gradients to 1e-4."""
    p, v = helpers.wide_image_scene()
    res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v)
    sb = f['screen_bounds'].astype(np.int64)
    assert ((sb[:, 0] // 16 >= 1024) & (f['n_touched'] > 0)).sum() > 100 and ((sb[:, 0] // 16 < 1024) & (f['n_touched'] > 0)).sum() > 100
    gi = np.random.default_rng(2).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    g = oracle.backward(f, S, gi)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                 dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, g, truth=oracle.forward_backward_f64(f, S, gi))


def test_more_big_footprints_than_a_sort_workgroup_collects(hip_backend, oracle):
    """~450 footprints of more than 256 candidate tiles inside one workgroup of the depth sort's last pass (it stages 256 in LDS, the rest append
    directly): forward intermediates bit-exact, gradients to 1e-4 (hot-accumulator replicas included: every one of them is a hot Gaussian)."""
    p, v = helpers.many_big_footprints_scene()
    # 450 faint screen-sized Gaussians: their alpha = 1/255 contours cross ~155 of the 130 k pixels inside the oracle's ULP band (1.2e-3 of the image)
    res, f, dp, RS, S = _forward_check(hip_backend, oracle, p, v, max_masked_pixels=3e-3)
    gi = np.random.default_rng(6).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    g = oracle.backward(f, S, gi)
    grads = hip_backend.backward(torch.empty(0, device=DEV), torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'],
                                 dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    _grads_close(grads, g, truth=oracle.forward_backward_f64(f, S, gi))


@pytest.mark.parametrize('near,far,zscale,passes', helpers.DEPTH_RANGE_CASES)
def test_depth_sort_pass_counts_through_the_forward(hip_backend, oracle, near, far, zscale, passes):
    """4, 3, 2 and 1 passes of the depth sort through the whole forward pass on hardware (helpers.depth_range_scene): counts, sorted keys, ranges and tile
    keys exactly, every tile's list as a set in depth order (tied keys -- all 800 of them in the one-pass case -- keep K1's compaction order)."""
    p, v = helpers.depth_range_scene(near, far, zscale)
    S, RS = helpers.settings_pair(v, device=DEV)
    dp = _to(p)
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    torch.cuda.synchronize()
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    helpers.check_lists_up_to_ties(helpers.decode_forward(hip_backend, res, 800, v.width, v.height), f)
    if len(np.unique(f['depth_keys'])) == f['V']:                                      # no ties: the image is the oracle's
        assert helpers.rel_inf(res.image.cpu().numpy(), f['image']) < 1e-4


def test_public_operators_autograd_and_fused_adam(hip_backend, oracle):
    """diff_rasterize -> loss.backward() -> FusedAdam.step(): the call sequence of Trainer.py:170-199."""
    from FasterGSCudaBackend import FusedAdam, diff_rasterize
    params, view = make_s0()
    S, RS = helpers.settings_pair(view, device=DEV)
    P = [torch.nn.Parameter(params[k].to(DEV)) for k in helpers.NAMES]
    dens = torch.zeros(2, 1000, device=DEV)
    image = diff_rasterize(*P, dens, RS)
    gi = torch.randn(image.shape, generator=torch.Generator().manual_seed(0))
    (image * gi.to(DEV)).sum().backward()
    f = oracle.forward(*helpers.np_params(params), S)
    dens_o = np.zeros((2, 1000), np.float32)
    g = oracle.backward(f, S, gi.numpy(), dens_o)
    _grads_close([p.grad for p in P], g, truth=oracle.forward_backward_f64(f, S, gi.numpy()))
    assert helpers.rel_inf(dens.cpu().numpy(), dens_o) < 1e-4
    lrs = [1.6e-4, 5e-3, 1e-3, 2.5e-2, 2.5e-3, 1.25e-4]
    opt = FusedAdam([{'params': [p], 'lr': lr} for p, lr in zip(P, lrs)], lr=0.0, eps=1e-15)
    ref = [(params[k].numpy().copy(), np.zeros(params[k].shape, np.float32), np.zeros(params[k].shape, np.float32)) for k in helpers.NAMES]
    grads_np = [p.grad.cpu().numpy().copy() for p in P]
    for step in (1, 2, 3):
        opt.step()
        for (pp, m, v), gg, lr in zip(ref, grads_np, lrs):
            oracle.adam_step(np.ascontiguousarray(gg), pp, m, v, step, lr)
    for p, (pp, m, v) in zip(P, ref):
        assert helpers.rel_inf(p.detach().cpu().numpy(), pp) < 1e-6
        st = opt.state[p]
        assert helpers.rel_inf(st['exp_avg'].cpu().numpy(), m) < 1e-6 and helpers.rel_inf(st['exp_avg_sq'].cpu().numpy(), v) < 1e-6
    # parameters without .grad are skipped (adam.py:16)
    opt.zero_grad()
    before = P[0].detach().clone()
    opt.step()
    assert torch.equal(before, P[0].detach())


@pytest.mark.parametrize('to_chw,clamp', [(True, True), (False, True), (False, False)])
def test_inference_variants(hip_backend, oracle, to_chw, clamp):
    from FasterGSCudaBackend import rasterize
    params, view = make_s0()
    params['sh_coefficients_0'] = params['sh_coefficients_0'] * 3.0
    S, RS = helpers.settings_pair(view, bg=(0.3, 0.1, 0.9), device=DEV)
    dp = _to(params)
    img = rasterize(*[dp[k] for k in helpers.NAMES], RS, to_chw, clamp)
    f = oracle.forward(*helpers.np_params(params), S, inference=True, to_chw=to_chw, clamp_output=clamp)
    assert img.shape == f['image'].shape
    assert helpers.rel_inf(img.cpu().numpy(), f['image']) < 1e-4


def test_fused_backward_adam_matches_unfused(hip_backend):
    """SURVEY.md D3. Not bit-exact on hardware: the two backward passes order their float atomics differently."""
    params, view = make_s0(n=2000)
    params['means'][:100, 2] = -10.0
    _, RS = helpers.settings_pair(view, device=DEV)
    order = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3]
    ref_p = {k: params[k].to(DEV).clone() for k in order}
    # seeded, with a floor under exp_avg_sq (helpers.seeded_moments): unseeded moments made this test depend on the global RNG state the tests before it
    # left behind -- an exp_avg_sq entry near zero turns the step into lr * m / (0.03 |g|), whose relative error is that of a tiny gradient (round 6:
    # one failure in five suite runs, 6e-4 on one element, none in isolation)
    seeded = {k: helpers.seeded_moments(params[k].shape, 71 + i) for i, k in enumerate(order)}
    ref_m = {k: seeded[k][0].to(DEV) for k in order}
    ref_v = {k: seeded[k][1].to(DEV) for k in order}
    fus_p, fus_m, fus_v = ({k: d[k].clone() for k in order} for d in (ref_p, ref_m, ref_v))
    gi = torch.randn(3, view.height, view.width, generator=torch.Generator().manual_seed(2)).to(DEV)
    dens_ref, dens_fus = torch.zeros(2, 2000, device=DEV), torch.zeros(2, 2000, device=DEV)
    start = {k: ref_p[k].clone() for k in order}
    for step in (1, 2):
        res = hip_backend.forward(*[ref_p[k] for k in helpers.NAMES], RS)
        grads = hip_backend.backward(dens_ref, gi, res.image, ref_p['means'], ref_p['scales'], ref_p['rotations'], ref_p['opacities'],
                                     ref_p['sh_coefficients_rest'], res.buffers, RS, res.state)
        gmap = dict(zip(helpers.NAMES, grads))
        hip_backend.adam_step_multi([gmap[k] for k in order], [ref_p[k] for k in order], [ref_m[k] for k in order],
                                    [ref_v[k] for k in order], [step] * 6, lrs, 0.9, 0.999, 1e-15)
        res2 = hip_backend.forward(*[fus_p[k] for k in helpers.NAMES], RS)
        hip_backend.backward_adam_fused(dens_fus, gi, res2.image, [fus_p[k] for k in order], [fus_m[k] for k in order],
                                        [fus_v[k] for k in order], res2.buffers, RS, res2.state, step, lrs)
    for k in order:
        delta_ref, delta_fus = (ref_p[k] - start[k]).cpu().numpy(), (fus_p[k] - start[k]).cpu().numpy()
        assert np.abs(delta_ref).max() > 0
        assert helpers.rel_inf(delta_fus, delta_ref) < 1e-4, k          # measured 1.2e-5 (float atomics in another order)
        assert helpers.rel_inf(fus_m[k].cpu().numpy(), ref_m[k].cpu().numpy()) < 1e-4, k
    assert helpers.rel_inf(dens_fus.cpu().numpy(), dens_ref.cpu().numpy()) < 1e-4


def test_empty_scene(hip_backend):
    params, view = make_s0(n=16)
    _, RS = helpers.settings_pair(view, bg=(0.1, 0.2, 0.3), device=DEV)
    empty = {k: v[:0].contiguous().to(DEV) for k, v in params.items()}
    res = hip_backend.forward(*[empty[k] for k in helpers.NAMES], RS)
    assert res.state[:2] == (0, 0)
    assert torch.allclose(res.image.cpu(), torch.tensor([0.1, 0.2, 0.3])[:, None, None].expand(3, 128, 128))


def _flip_aware_forward_backward(hip_backend, oracle, params, view, label, adam_steps=0, K=16, aa=False, max_masked=1e-3, near_tol=None,
                                 last_contributor_budget=1e-4, image_flip_budget=0):
    """Forward + backward (+ FusedAdam-style steps with the same gradients) against the oracle; entries on a hard threshold are
    counted and excluded (helpers.check_flip_aware), everything else is held to 1e-4 -- image, six gradients, densification_info,
    and after `adam_steps` Adam steps the parameters and both moments."""
    S, RS = helpers.settings_pair(view, K, aa, device=DEV)
    dp = _to(params)
    n = dp['means'].shape[0]
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    f = oracle.forward(*helpers.np_params(params), S, bucket_size=64)
    dec = helpers.decode_forward(hip_backend, res, n, view.width, view.height)
    assert abs(dec['V'] - f['V']) <= max(2, n // 1000) and abs(dec['I'] - f['I']) <= max(16, n // 100)
    masks = helpers.flip_masks(oracle, f, S, dec)
    gi = np.random.default_rng(3).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    dens_o = np.zeros((2, n), np.float32)
    g = oracle.backward(f, S, gi, dens_o)
    dens = torch.zeros(2, n, device=DEV)
    grads = hip_backend.backward(dens, torch.from_numpy(gi).to(DEV), res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'],
                                 dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    got = {k: t.cpu().numpy() for k, t in zip(helpers.GRAD_KEYS, grads)}
    got['densification_info'] = dens.cpu().numpy().T
    ref = {k: g[k] for k in helpers.GRAD_KEYS}
    ref['densification_info'] = dens_o.T
    truth = oracle.forward_backward_f64(f, S, gi)              # the same formulas in double: the element-wise bar is applied three-way
    report = helpers.check_flip_aware(res.image.cpu().numpy(), f['image'], got, ref, masks, max_masked=max_masked, label=label, truth=truth, near_tol=near_tol, image_flip_budget=image_flip_budget)
    # integer intermediates away from the thresholds: the pixel's last contributor
    npr = helpers.tiles_to_image(dec['n_processed_tiles'], view.width, view.height)
    if dec['I'] == f['I']:
        assert (npr != f['n_processed'].reshape(npr.shape))[~masks['pixel']].mean() < last_contributor_budget
    if adam_steps:
        order = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
        gkey = dict(zip(helpers.NAMES, helpers.GRAD_KEYS))
        lrs = [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3]
        gmap = dict(zip(helpers.NAMES, grads))
        P = [dp[k].clone() for k in order]
        M, V = [torch.zeros_like(t) for t in P], [torch.zeros_like(t) for t in P]
        oP = [np.ascontiguousarray(params[k].numpy().copy()) for k in order]
        oM, oV = [np.zeros_like(t) for t in oP], [np.zeros_like(t) for t in oP]
        grads_np = [np.ascontiguousarray(gmap[k].cpu().numpy().reshape(t.shape)) for k, t in zip(order, oP)]   # the same gradient on both sides
        for step in range(1, adam_steps + 1):
            hip_backend.adam_step_multi([gmap[k] for k in order], P, M, V, [step] * 6, lrs, 0.9, 0.999, 1e-15)
            for i, lr in enumerate(lrs):
                oracle.adam_step(grads_np[i], oP[i], oM[i], oV[i], step, lr)
        for i, k in enumerate(order):
            assert helpers.rel_inf(P[i].cpu().numpy(), oP[i]) < 1e-6, (label, k)
            assert helpers.rel_inf(M[i].cpu().numpy(), oM[i]) < 1e-6 and helpers.rel_inf(V[i].cpu().numpy(), oV[i]) < 1e-6, (label, k)
    return report


def test_mid_size_against_oracle(hip_backend, oracle):
    """60 k garden-like Gaussians at 640x360: every output to 1e-4 outside the counted threshold mask, 3 Adam steps."""
    params = make_garden_like(60_000)
    params['scales'] = params['scales'] + 0.7          # keep footprints comparable to 1080p statistics at this resolution
    v = orbit_views(8, width=640, height=360, focal=473.0)[1]
    _flip_aware_forward_backward(hip_backend, oracle, params, v, '60k', adam_steps=3)


@pytest.mark.parametrize('scene,n,view', [('S1', 1_000_000, 0), ('S2', 3_000_000, 3)])
def test_full_size_against_oracle(hip_backend, oracle, scene, n, view):
    """BASELINE.json full sizes (S1 = 1 M, S2 = 3 M Gaussians, the headline workload, at 1920x1080) against the oracle itself:
    image, six gradients, densification_info to 1e-4 outside the counted threshold mask, then 3 Adam steps on all 59 N floats."""
    params = make_garden_like(n)
    _flip_aware_forward_backward(hip_backend, oracle, params, orbit_views(8)[view], scene, adam_steps=3)


@pytest.mark.parametrize('scene,n,view,aa', [('S1', 1_000_000, 0, False), ('S2', 3_000_000, 3, False), ('S2 proper AA', 3_000_000, 6, True)])
def test_full_size_records_against_fp64_conditioning_aware(hip_backend, oracle, scene, n, view, aa):
    """K1 on its own (round 6; helpers.check_records_against_f64): every record component at most 4 x as far from its fp64 value as the fp32 oracle's is
    (+ 2e-6): a needle's conic may be 2e-3 apart between the two fp32 runs exactly when the reference arithmetic itself is that far from the truth."""
    r = helpers.check_records_against_f64(hip_backend, oracle, make_garden_like(n), orbit_views(8)[view], aa=aa, device=DEV, label=scene)
    assert r['visible'] > n // 10


@pytest.mark.parametrize('scene,n,view,shift', [('S2', 3_000_000, 3, 0.0), ('S2 layered', 3_000_000, 3, -3.0)])
def test_full_size_blend_kernels_on_the_device_records(hip_backend, oracle, scene, n, view, shift):
    """K10 / K11 isolated from K1 at the headline size (round 6; helpers.check_blend_on_device_records): the oracle's blend re-run on the records the device's K1
    produced, over the same instance lists -- image, final transmittance, last contributors and K11's nine per-Gaussian sums to 1e-4 outside the risk masks."""
    params = make_garden_like(n)
    params['opacities'] = params['opacities'] + shift
    helpers.check_blend_on_device_records(hip_backend, oracle, params, orbit_views(8)[view], device=DEV, label=scene, max_masked=3e-3)


def test_layered_scene_full_size_against_oracle(hip_backend, oracle):
    """bench.py's `layered_scene` at the size it is timed (missing #2): S2 with every opacity logit lowered by 3 -- ~11 of the
    ~13 buckets of a tile are blended instead of 2 of 21, the regime of a trained scene, where K10 / K11 are more than half of the step.
    Forward, backward, densification_info and 3 Adam steps against the oracle (kernels_backward.cuh:286-471, adam.cu:10-34). The
    threshold-risk masks grow with the number of (pixel, Gaussian) pairs per pixel: bounded at 3e-3 here (1e-3 for S2)."""
    params = make_garden_like(3_000_000)
    params['opacities'] = params['opacities'] - 3.0
    _flip_aware_forward_backward(hip_backend, oracle, params, orbit_views(8)[3], 'S2 layered', adam_steps=3, max_masked=3e-3)


def test_s3_six_million_against_oracle(hip_backend, oracle):
    """S3 (6 M Gaussians, the top of north_star's '~1-6 M' range; profiles/r0x_s3_bench_line.json times it): forward + backward +
    densification_info against the oracle, flip-aware, max-norm and element-wise 1e-4."""
    params = make_garden_like(6_000_000)
    _flip_aware_forward_backward(hip_backend, oracle, params, orbit_views(8)[5], 'S3', adam_steps=0)


@pytest.mark.parametrize('K,aa', [(4, False), (16, True)])
def test_s1_sh_degree_one_and_proper_antialiasing_against_oracle(hip_backend, oracle, K, aa):
    """active_sh_bases < 16 and proper_antialiasing=True at a BASELINE size (S1 = 1 M Gaussians, 1920x1080), not only on the
    3e5-Gaussian fuzz scenes: forward, six gradients, densification_info flip-aware to 1e-4 (kf:148-155 the opacity compensation, sh:14-68 the
    degree cut; kb:15-257 their derivatives)."""
    params = make_garden_like(1_000_000)
    _flip_aware_forward_backward(hip_backend, oracle, params, orbit_views(8)[2], f'S1 K={K} aa={int(aa)}', adam_steps=0, K=K, aa=aa)


@pytest.mark.parametrize('to_chw,clamp', [(True, True), (False, True), (False, False)])
def test_inference_at_s2_against_oracle(hip_backend, oracle, to_chw, clamp):
    """VERDICT r4 missing #4: the inference kernels (ki:14-207 colour clamped at store, no n_touched clearing; ki:348-463 CHW / HWC epilogue,
    optional output clamp) against oracle.forward(inference=True) at the size bench.py's render leg times (S2 = 3 M Gaussians, 1080p), not only at
    S0 size. Flip-aware: the pixels that own a (pixel, Gaussian) pair on the alpha / transmittance thresholds come from the oracle's training
    forward of the same scene (the tests depend on geometry and opacity only) and are counted and excluded; everything else to 1e-4."""
    from FasterGSCudaBackend import rasterize
    params = make_garden_like(3_000_000)
    params['sh_coefficients_0'] = params['sh_coefficients_0'] * 2.0          # push colours beyond [0, 1] on both sides: the colour clamp at
    params['sh_coefficients_0'][::3] += 4.0                                  # store (ki:200) and the output clamp (ki:445-449) must both matter
    view = orbit_views(8)[3]
    S, RS = helpers.settings_pair(view, bg=(0.3, 0.1, 0.9), device=DEV)
    dp = _to(params)
    img = rasterize(*[dp[k] for k in helpers.NAMES], RS, to_chw, clamp).cpu().numpy()
    f_inf = oracle.forward(*helpers.np_params(params), S, inference=True, to_chw=to_chw, clamp_output=clamp)
    f_train = oracle.forward(*helpers.np_params(params), S, bucket_size=64)
    pm = helpers.flip_masks(oracle, f_train, S)['pixel']
    assert img.shape == f_inf['image'].shape
    chw = lambda x: x if to_chw else np.moveaxis(x, -1, 0)
    err = np.abs(chw(img).astype(np.float64) - chw(f_inf['image'])).max(axis=0)
    scale = max(1.0, float(np.abs(f_inf['image']).max()))
    helpers.log_note('inference_s2', f'{float(err[~pm].max() / scale):.3e}', to_chw=int(to_chw), clamp=int(clamp), masked=f'{float(pm.mean()):.3e}',
                     masked_max=f'{float(err[pm].max() / scale) if pm.any() else 0.0:.3e}')
    assert float(pm.mean()) < 1e-3 and float(err[~pm].max()) < 1e-4 * scale, (float(pm.mean()), float(err[~pm].max()), scale)
    assert not pm.any() or float(err[pm].max()) < 5e-2 * scale
    if clamp:
        assert float(img.max()) <= 1.0 and float(img.min()) >= 0.0
    else:
        assert float(f_inf['image'].max()) > 1.0                               # the scene does exceed 1 where nothing clamps the output


@pytest.mark.parametrize('label,n,width,height,focal,max_masked,flips', [('S1 at 4K', 1_000_000, 3840, 2160, 2840.0, 4e-3, 0),
                                                                          ('300k at 8K (32-bit tile keys)', 300_000, 7680, 4320, 5680.0, 2e-2, 4)])
def test_large_images_against_oracle(hip_backend, oracle, label, n, width, height, focal, max_masked, flips):
    """Sizes beyond the benchmark's 1080p: 3840 x 2160 (43 200 tiles: 16-bit keys sorted in two 8-bit passes, four times the instances per Gaussian) and
    7680 x 4320 (172 800 tiles: the 32-bit key path of fwd:152-153 at scale, 18 bits in three passes) -- forward, six gradients and densification_info
    flip-aware to 1e-4, the instance count against the oracle's. The share of Gaussians that own a (pixel, Gaussian) pair inside the oracle's ULP band
    around the alpha threshold grows with the pixels a Gaussian covers (4.3e-4 at 1080p, 1.6e-3 at 4K: the contour AND the band widen), so the mask
    bound is scaled with the image. At 8K up to four of the 33 M pixels may hold an alpha-test flip outside the oracle's band (measured: one, 2.8e-3 --
    identical tile lists, tools/archive/diag_8k.py): at pixel coordinates of several thousand the exponent's three terms are ~1e1 with 1e-6 of rounding each,
    and the HIP blend contracts them into FMAs."""
    params = make_garden_like(n)
    view = orbit_views(8, width=width, height=height, focal=focal)[4]
    _flip_aware_forward_backward(hip_backend, oracle, params, view, label, adam_steps=0, max_masked=max_masked, image_flip_budget=flips)


def test_twenty_million_gaussians_forward_properties(hip_backend):
    """Beyond north_star's '~1-6 M' range: 20 M Gaussians at 1080p (13 M visible, ~10^8 instances; the footprint rows, their sums per wave segment / block
    and the big-footprint list at their largest sizes in the suite). No oracle run at this size: the size-independent properties -- checksum of checksums,
    sorted keys, consistent ranges and bucket offsets, depth order inside sampled tiles, a finite image whose final transmittance lies in [0, 1]."""
    n = 20_000_000
    params = make_garden_like(n)
    v = orbit_views(8)[6]
    _, RS = helpers.settings_pair(v, device=DEV)
    dp = _to(params)
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(hip_backend, res, n, v.width, v.height)
    assert dec['V'] == int((dec['n_touched'] > 0).sum()) and dec['I'] == int(dec['n_touched'].sum()) and dec['V'] > n // 2
    sel = 1 if np.all(np.diff(dec['depth_keys1'].astype(np.int64)) >= 0) else 0                         # which half holds the depth-sorted list
    assert np.all(np.diff(dec[f'depth_keys{sel}'].astype(np.int64)) >= 0)
    counts = dec['n_touched'][dec[f'prim_idx{sel}']].astype(np.int64)                                   # tile counts in depth order
    assert np.array_equal(dec['offsets'].astype(np.int64), np.cumsum(counts) - counts)                  # K5's own offsets = their exclusive prefix
    keys = dec['inst_keys'].astype(np.int64)
    assert np.all(np.diff(keys) >= 0)
    ranges = dec['ranges'].astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    assert lens.sum() == dec['I'] and ranges[:, 1].max() == dec['I'] and np.array_equal(np.cumsum((lens + 63) // 64), dec['bucket_offsets'].astype(np.int64))
    assert np.all(keys == np.repeat(np.arange(len(ranges)), lens))
    w2c = v.w2c.numpy()
    depth = params['means'].numpy() @ w2c[2, :3] + w2c[2, 3]
    for t in np.random.default_rng(1).choice(len(ranges), 100, replace=False):
        # (lists of 12 k entries: neighbours are ~1e-4 apart, numpy's depth and the kernel's differ in the last bit -- hence the 2e-6)
        assert np.all(np.diff(depth[dec['inst_prims'][ranges[t, 0]:ranges[t, 1]]].astype(np.float64)) >= -2e-6)
    assert torch.isfinite(res.image).all() and float(res.image.min()) >= 0.0
    fT = helpers.tiles_to_image(dec['final_T_tiles'], v.width, v.height)
    assert fT.min() >= 0.0 and fT.max() <= 1.0


def test_full_size_properties(hip_backend):
    """BASELINE.json full size (1920x1080, 1 M Gaussians): size-independent properties instead of an oracle run."""
    params = make_garden_like(1_000_000)
    v = orbit_views(8)[0]
    _, RS = helpers.settings_pair(v, device=DEV)
    dp = _to(params)
    n = 1_000_000
    res = hip_backend.forward(*[dp[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(hip_backend, res, n, v.width, v.height)
    assert dec['V'] == int((dec['n_touched'] > 0).sum()) and dec['I'] == int(dec['n_touched'].sum())     # checksum of checksums
    keys = dec['inst_keys'].astype(np.int64)
    assert np.all(np.diff(keys) >= 0)                                                                     # tile keys sorted
    ranges = dec['ranges'].astype(np.int64)
    assert ranges[:, 1].max() == dec['I'] and np.all(ranges[:, 1] >= ranges[:, 0])
    lens = ranges[:, 1] - ranges[:, 0]
    assert lens.sum() == dec['I'] and np.array_equal(np.cumsum((lens + 63) // 64), dec['bucket_offsets'].astype(np.int64))
    w2c = v.w2c.numpy()
    depth = params['means'].numpy() @ w2c[2, :3] + w2c[2, 3]
    rng = np.random.default_rng(0)
    for t in rng.choice(len(ranges), 200, replace=False):                                                 # depth-sorted inside tiles
        d = depth[dec['inst_prims'][ranges[t, 0]:ranges[t, 1]]]
        assert np.all(np.diff(d) >= 0)
    assert np.all(keys == np.repeat(np.arange(len(ranges)), lens))
    img = res.image
    assert torch.isfinite(img).all() and float(img.min()) >= 0.0
    fT = helpers.tiles_to_image(dec['final_T_tiles'], v.width, v.height)
    assert fT.min() >= 0.0 and fT.max() <= 1.0
    # inference path renders the same picture (clamped)
    img2 = hip_backend.inference(*[dp[k] for k in helpers.NAMES], RS, True, True)
    assert float((img2 - img.clamp(0, 1)).abs().max()) < 1e-5
    # backward is linear in grad_image: grads(2g) == 2 grads(g)   (atomics order -> tolerance)
    gi = torch.randn(img.shape, generator=torch.Generator().manual_seed(1)).to(DEV) / img.numel()
    args = (res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    g1 = hip_backend.backward(torch.empty(0, device=DEV), gi, *args)
    g2 = hip_backend.backward(torch.empty(0, device=DEV), 2.0 * gi, *args)
    for a, b in zip(g1, g2):
        assert torch.isfinite(a).all()
        assert float((2.0 * a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12
    # gradients of invisible Gaussians are exactly zero
    invisible = torch.from_numpy(dec['n_touched'] == 0).to(DEV)
    for a in g1:
        assert float(a[invisible].abs().max()) == 0.0


def test_async_forward_through_the_public_operators(hip_backend, oracle):
    """set_async_forward(True): after one synchronous pass PER VIEW the training forward issues no host wait (fgs_forward_async, capacity from
    that view's instances-per-Gaussian ratio); image and gradients stay those of the synchronous path. Passes without gradients are checked
    synchronously. With a headroom below 1 the capacity is exceeded: backward notices (RuntimeWarning) and returns zero gradients."""
    import warnings
    from FasterGSCudaBackend import async_forward_stats, diff_rasterize, set_async_forward
    params, view = make_s0(n=3000)
    S, RS = helpers.settings_pair(view, device=DEV)
    f = oracle.forward(*helpers.np_params(params), S)
    gi = torch.randn(3, view.height, view.width, generator=torch.Generator().manual_seed(0))
    g = oracle.backward(f, S, gi.numpy())

    def step():
        P = [torch.nn.Parameter(params[k].to(DEV)) for k in helpers.NAMES]
        image = diff_rasterize(*P, torch.empty(0, device=DEV), RS)
        (image * gi.to(DEV)).sum().backward()
        return image.detach(), [p.grad for p in P]
    set_async_forward(False)
    ref_image, _ = step()
    try:
        set_async_forward(True)
        step()                                   # first visit of this view: synchronous, records the view's ratio
        st = async_forward_stats()
        assert st['ratio'] > 0 and st['views'] == 1
        for _ in range(2):
            image, grads = step()                # asynchronous
            assert torch.equal(image, ref_image)
            _grads_close(grads, g)
        assert async_forward_stats()['overflows'] == 0
        # a pass that needs no gradient is checked synchronously: the image is complete whatever the recorded ratio says
        set_async_forward(True, headroom=0.4)
        with torch.no_grad():
            img = diff_rasterize(*[params[k].to(DEV) for k in helpers.NAMES], torch.empty(0, device=DEV), RS)
        assert torch.equal(img, ref_image) and async_forward_stats()['overflows'] == 0
        # capacity < need (headroom 0.4): backward notices, warns and returns ZERO gradients -- never a gradient of a truncated image
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            _, grads = step()
        assert any('exceeded the capacity' in str(x.message) for x in w) and async_forward_stats()['overflows'] == 1
        assert all(float(t.abs().max()) == 0.0 for t in grads)
        # ... and the optimizer step that would consume those zeros is SKIPPED: no step count, no motion on momentum (round-3 advisor finding) -- by the
        # optimizer that OWNS the parameters of the overflowed pass, and only by it (round-4 advisor finding: the mark names those parameters)
        from FasterGSCudaBackend import FusedAdam, take_async_overflow
        assert take_async_overflow()                                             # the mark of the pass above (its parameters are gone): taken here
        P = [torch.nn.Parameter(params[k].to(DEV)) for k in helpers.NAMES]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (diff_rasterize(*P, torch.empty(0, device=DEV), RS) * gi.to(DEV)).sum().backward()      # overflows again (headroom 0.4): zero gradients, mark set
        q = torch.nn.Parameter(torch.ones(64, 3, device=DEV))
        other = FusedAdam([{'params': [q], 'lr': 1e-2}], lr=0.0, eps=1e-15)
        q.grad = torch.ones_like(q)
        other.step()                                                             # an unrelated optimizer neither skips nor consumes the mark
        assert float((q.detach() - 1.0).abs().max()) > 0.0 and other.state[q]['step'] == 1
        owner = FusedAdam([{'params': [p], 'lr': 1e-2} for p in P], lr=0.0, eps=1e-15)
        before = [p.detach().clone() for p in P]
        owner.step()                                                             # the owner consumes the mark: nothing happens
        assert all(torch.equal(a, p.detach()) for a, p in zip(before, P)) and not owner.state[P[0]] and not take_async_overflow()
        for p in P:
            p.grad = torch.ones_like(p)
        owner.step()                                                             # the next step is an ordinary one
        assert float((P[0].detach() - before[0]).abs().max()) > 0.0 and owner.state[P[0]]['step'] == 1
        # a mark nobody took is dropped by the next forward pass over the same parameters (it must not skip a later, valid step)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (diff_rasterize(*P, torch.empty(0, device=DEV), RS) * gi.to(DEV)).sum().backward()
        set_async_forward(True)
        with torch.no_grad():
            diff_rasterize(*P, torch.empty(0, device=DEV), RS)
        assert not take_async_overflow()
        set_async_forward(True, headroom=0.4)
        set_async_forward(True)                  # default headroom again: the refreshed ratio renders the view completely
        image, grads = step()
        assert torch.equal(image, ref_image)
        _grads_close(grads, g)
        # a pass that asks for gradients but never runs backward (an evaluation loop without no_grad) is checked by the next forward pass
        set_async_forward(True, headroom=0.4)
        P = [torch.nn.Parameter(params[k].to(DEV)) for k in helpers.NAMES]
        truncated = diff_rasterize(*P, torch.empty(0, device=DEV), RS)
        assert not torch.equal(truncated, ref_image) and async_forward_stats()['unchecked_passes'] == 1
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            with torch.no_grad():
                diff_rasterize(*[params[k].to(DEV) for k in helpers.NAMES], torch.empty(0, device=DEV), RS)
        assert any('backward never ran' in str(x.message) for x in w) and async_forward_stats()['unchecked_passes'] == 0
        set_async_forward(True)
        # a w2c edited in place is a NEW view (version counter), as is another tensor at a recycled address: both start synchronously
        RS.w2c.mul_(1.0)
        before = async_forward_stats()['views']
        image, grads = step()
        assert torch.equal(image, ref_image) and async_forward_stats()['views'] == before
        _grads_close(grads, g)
        # another view (another w2c tensor) starts with its own synchronous pass
        RS2 = RS._replace(w2c=RS.w2c.clone())
        P = [torch.nn.Parameter(params[k].to(DEV)) for k in helpers.NAMES]
        assert torch.equal(diff_rasterize(*P, torch.empty(0, device=DEV), RS2), ref_image) and async_forward_stats()['views'] == 2
    finally:
        set_async_forward(False)
