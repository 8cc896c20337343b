"""CPU tests: the product's real kernel sources + C-ABI host code + ctypes glue, compiled for the fiber simulator
(tests/sim), compared with the oracle on every intermediate. Catches indexing / masking / synchronisation / orchestration
bugs without a GPU. The simulator shares libm with the oracle and is built with -ffp-contract=off, so almost everything
is compared bit-exactly; gradients differ only by summation order."""
import numpy as np
import pytest
import torch

import helpers
from harness.scenes import View, make_s0


def _run(sim_backend, oracle, params, view, K=16, aa=False, bg=None, check_grads=True):
    S, RS = helpers.settings_pair(view, K, aa, bg)
    a = helpers.np_params(params)
    n = a[0].shape[0]
    res = sim_backend.forward(*[params[k] for k in helpers.NAMES], RS)
    f = oracle.forward(*a, S, bucket_size=64)
    dec = helpers.decode_forward(sim_backend, res, n, view.width, view.height)
    helpers.check_forward_against_oracle(dec, f, True, view.width, view.height, res.image.numpy())
    if f['B'] > 0:
        live = np.zeros((f['B'], 192), bool)          # checkpoints of pixels that were not done at that bucket are exact
        assert dec['B'] == f['B']
    if not check_grads:
        return res, f
    gi = np.random.default_rng(5).standard_normal(f['image'].shape).astype(np.float32)
    dens_o = np.zeros((2, n), np.float32)
    g = oracle.backward(f, S, gi, dens_o)
    dens = torch.zeros(2, n)
    grads = sim_backend.backward(dens, torch.from_numpy(gi), res.image, params['means'], params['scales'], params['rotations'],
                                 params['opacities'], params['sh_coefficients_rest'], res.buffers, RS, res.state)
    for k, t in zip(helpers.GRAD_KEYS, grads):
        assert helpers.rel_inf(t.numpy().reshape(g[k].shape), g[k]) < 1e-5, k
    assert helpers.rel_inf(dens.numpy(), dens_o) < 1e-5
    return res, f


@pytest.mark.parametrize('w,h,n', [(333, 211, 300), (16, 12, 40)])
def test_tile_plan_covers_every_tile_and_balances_the_xcds(sim_backend, w, h, n):
    """K10's device-side tile -> workgroup plan (binning.hip: plan_tiles_kernel): whatever the image size and the distribution of the
    Gaussians, every tile is blended by exactly one workgroup, every XCD gets ten blocks in descending weight, and the greedy deal keeps
    the heaviest XCD within one block of the mean. Training and inference (which plans from the same scan)."""
    p, v = make_s0(seed=5, n=n)
    p['means'][:, 1] = p['means'][:, 1].abs()                  # everything in the lower half of the image: a strong vertical work gradient
    v = View(v.w2c, v.position, w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, 0.2, 1e4, torch.zeros(3))
    _, RS = helpers.settings_pair(v)
    gw, gh = (w + 15) // 16, (h + 11) // 12
    ref = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)                 # default mapping (closed-form columns)
    assert sim_backend.lib.fgs_debug_set_option(10, 254) == 0                     # the block plan is an A/B option since the column mapping won
    try:
        res = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)
        dec = helpers.decode_forward(sim_backend, res, n, w, h)
        info = helpers.check_tile_plan(dec['tile_plan'], dec['bucket_offsets'], gw, gh)
        assert info['weights'].sum() == int(dec['bucket_offsets'][gw * gh - 1]) + gw * gh
        assert torch.equal(res.image, ref.image)                                  # the mapping never changes a result
        inf = sim_backend.inference(*[p[k] for k in helpers.NAMES], RS, True, True, return_state=True)
        lt = sim_backend.blob_layout(1, n, w, h, inf.state[1], inf.state[2])
        plan = sim_backend.view(inf.buffers[1], lt, 'tile_plan', torch.int32).numpy().view(np.uint32)
        assert np.array_equal(plan, dec['tile_plan'])
        for m in (0, 251):                                                        # bands, columns bottom-up (252 = the default above; row groups: test_gpu)
            assert sim_backend.lib.fgs_debug_set_option(10, m) == 0
            assert torch.equal(sim_backend.forward(*[p[k] for k in helpers.NAMES], RS).image, ref.image), m
    finally:
        sim_backend.lib.fgs_debug_set_option(10, 252)


def test_wave_primitives_selftest(sim_backend):
    out = torch.zeros(256, dtype=torch.int32)
    assert sim_backend.lib.fgs_debug_wave_selftest(out.data_ptr(), None) == 0
    o, l = out.numpy(), np.arange(64)
    assert np.array_equal(o[:64], np.where(l == 0, 1000, 99 + l)) and np.array_equal(o[64:128], 1000 + (l + 1) % 64)
    assert np.array_equal(o[128:192], (l + 2) // 3) and np.all(o[192:] == 2142)


def test_s0_all_intermediates(sim_backend, oracle):
    params, view = make_s0()
    _run(sim_backend, oracle, params, view)


def test_product_flavour_of_the_sources(sim_product_backend, sim_backend, oracle):
    """The other sim tests run the sources with -DFGS_DEV_SWITCHES (they compare formulations). libfgs_hip.so is built WITHOUT it: its switches are
    compile-time constants (csrc/fgs_kernels.h: FGS_SWITCH) and the A/B variants are not compiled. The same suite of checks on that flavour: every
    intermediate against the oracle, fused == backward -> Adam, Adam against the oracle, a synchronisation-free forward, and bit-identical
    results between the two flavours."""
    assert not hasattr(sim_product_backend.lib, 'fgs_debug_set_option') and hasattr(sim_backend.lib, 'fgs_debug_set_option')
    assert b'dev' not in sim_product_backend.lib.fgs_build_info() and b'dev' in sim_backend.lib.fgs_build_info()
    params, view = make_s0()
    res, _ = _run(sim_product_backend, oracle, params, view)
    params4, v = make_s0(seed=3, n=150)
    view4 = View(v.w2c, v.position, 130, 25, 0.8 * 130, 0.8 * 130, 65.0, 12.5, 0.2, 1e4, torch.tensor([0.2, 0.5, 0.7]))
    _run(sim_product_backend, oracle, params4, view4, K=9, aa=True)
    _, RS = helpers.settings_pair(view)
    dev = sim_backend.forward(*[params[k] for k in helpers.NAMES], RS)
    assert torch.equal(res.image, dev.image) and res.state == dev.state
    small, view_s = make_s0(n=400)
    small['means'][:40, 2] = -10.0
    fused_equals_backward_then_adam(sim_product_backend, small, view_s)
    test_adam_single_and_multi(sim_product_backend, oracle)
    test_forward_without_host_synchronisation(sim_product_backend, oracle)


@pytest.mark.parametrize('w,h,K,aa', [(48, 36, 4, True), (50, 30, 16, False), (16, 12, 1, False), (130, 25, 9, True)])
def test_partial_tiles_sh_degrees_antialiasing(sim_backend, oracle, w, h, K, aa):
    p, v = make_s0(seed=3, n=150)
    v = View(v.w2c, v.position, w, h, 0.8 * w, 0.8 * w, w / 2.0, h / 2.0, 0.2, 1e4, torch.tensor([0.2, 0.5, 0.7]))
    _run(sim_backend, oracle, p, v, K, aa)


def test_large_footprints_take_the_cooperative_path(sim_backend, oracle):
    """Gaussians covering > 4 candidate tiles exercise the wave-cooperative branches of preprocess / create_instances."""
    p, v = make_s0(seed=7, n=200)
    p['scales'] = p['scales'] + 2.3
    _, f = _run(sim_backend, oracle, p, v)
    assert (f['n_touched'] > 4).sum() > 50 and (f['n_touched'] > 68).sum() > 5


def test_huge_footprints_take_the_workgroup_path(sim_backend, oracle):
    """> 1024 candidate tiles: the instance generator hands the footprint to create_instances_big_kernel."""
    p, v = make_s0(seed=13, n=40)
    v = View(v.w2c, v.position, 640, 480, 500.0, 500.0, 320.0, 240.0, 0.2, 1e4, torch.zeros(3))
    p['scales'][:6] = p['scales'][:6] + 3.2          # six screen-filling Gaussians
    p['scales'][6:20] = p['scales'][6:20] + 1.8      # medium ones (33..1024 candidates)
    _, f = _run(sim_backend, oracle, p, v, check_grads=False)
    sb = f['screen_bounds'].astype(np.int64)
    n_max = ((sb[:, 1] + 15) // 16 - sb[:, 0] // 16) * ((sb[:, 3] + 11) // 12 - sb[:, 2] // 12)
    assert (n_max > 1024).sum() >= 3 and ((n_max > 32) & (n_max <= 1024)).sum() >= 3 and (n_max <= 32).sum() >= 3


def test_hot_footprints_accumulate_through_replicas(sim_backend, oracle):
    """Footprints above 256 candidate tiles get private accumulator replicas in K11 (fgs_config.h: kHotFootprint), folded into the
    planes afterwards: gradients must not notice. 25 x 25 tiles, a few Gaussians covering 300+ of them next to small ones."""
    p, v = make_s0(seed=13, n=24)
    v = View(v.w2c, v.position, 400, 300, 320.0, 320.0, 200.0, 150.0, 0.2, 1e4, torch.zeros(3))
    p['scales'][:5] = p['scales'][:5] + 2.6          # hot: hundreds of tiles each
    p['scales'][5:9] = p['scales'][5:9] + 1.5        # medium
    _, f = _run(sim_backend, oracle, p, v)
    sb = f['screen_bounds'].astype(np.int64)
    n_max = ((sb[:, 1] + 15) // 16 - sb[:, 0] // 16) * ((sb[:, 3] + 11) // 12 - sb[:, 2] // 12)
    assert ((n_max > 256) & (f['n_touched'] > 0)).sum() >= 3 and ((n_max <= 256) & (f['n_touched'] > 0)).sum() >= 5


def test_long_tile_lists_span_several_batches(sim_backend, oracle):
    """> 192 Gaussians per tile: multiple LDS batches and several buckets per tile."""
    p, v = make_s0(seed=11, n=1500)
    p['means'][:, :2] *= 0.15
    p['opacities'] -= 2.5
    _, f = _run(sim_backend, oracle, p, v)
    assert (f['ranges'][:, 1] - f['ranges'][:, 0]).max() > 400


@pytest.mark.parametrize('variant', [1, 4, 5])
def test_strip_backward_variant(sim_backend, oracle, variant):
    """The default is the systolic formulation; the lane = pixel ones (1: DPP reductions, 4: matrix-core reduction through an LDS transposition, the
    matrix instruction emulated as the fmaf chain it is) stay selectable and must give the same gradients."""
    sim_backend.lib.fgs_debug_set_backward_variant(variant)
    try:
        p, v = make_s0(seed=11, n=600)
        p['means'][:, :2] *= 0.3
        _run(sim_backend, oracle, p, v)
    finally:
        sim_backend.lib.fgs_debug_set_backward_variant(3)


@pytest.mark.parametrize('waves', [1, 3])
def test_chained_backward_variant_on_long_chains(sim_backend, oracle, waves):
    """K11 variant 5 (dev library exhibit, round 6): the items of a wave follow each other through the lanes without draining. With 1 / 3 waves
    (option 14) a wave chains 79 / 27 items: the descriptor window (32 lanes, refilled 16 at a time) wraps four times, items of every length follow
    each other. Same gradients as the oracle."""
    sim_backend.lib.fgs_debug_set_backward_variant(5)
    assert sim_backend.lib.fgs_debug_set_option(14, waves) == 0
    try:
        p, v = make_s0(seed=11, n=1500)
        p['means'][:, :2] *= 0.3
        p['opacities'] -= 2.0
        _run(sim_backend, oracle, p, v)
    finally:
        sim_backend.lib.fgs_debug_set_backward_variant(3)
        sim_backend.lib.fgs_debug_set_option(14, 4096)


def test_uninitialised_scratch_is_harmless(sim_backend, oracle):
    """Scratch buffers pre-filled with 0xFF (NaN): checkpoints of finished pixels, records of invisible primitives etc. are
    never written by the forward pass and must never leak into a result (regression: 0 * NaN in the branch-free K11)."""
    p, v = make_s0(seed=11, n=1500)
    p['means'][:, :2] *= 0.15
    p['opacities'] -= 2.5
    p['means'][:50, 2] = -10.0                      # invisible primitives: their records stay poisoned
    be = helpers.poisoned(sim_backend)
    for variant in (2, 3, 4, 5):                    # the default, round 1's main form, the matrix-core form, the chained form (0 / 1: on hardware, test_gpu_parity.py)
        be.lib.fgs_debug_set_backward_variant(variant)
        try:
            _run(be, oracle, p, v)
        finally:
            be.lib.fgs_debug_set_backward_variant(3)


def test_second_backward_over_the_same_buffers(sim_product_backend, oracle):
    """A retained graph differentiated twice: K11's accumulator records live in the forward pass's primitive blob and are cleared by K1 (round 6);
    the first backward pass leaves sums in them and a flag (counters[7]) that makes the second one clear them itself. Same gradients, bit for bit,
    also for another upstream gradient in between."""
    be = sim_product_backend
    p, v = make_s0(seed=5, n=700)
    p['scales'] = p['scales'] + 0.8                                   # some hot (large-footprint) Gaussians as well: their replicas are cleared per pass
    S, RS = helpers.settings_pair(v)
    res = be.forward(*[p[k] for k in helpers.NAMES], RS)
    gi = torch.randn(res.image.shape, generator=torch.Generator().manual_seed(2))
    args = (res.image, p['means'], p['scales'], p['rotations'], p['opacities'], p['sh_coefficients_rest'], res.buffers, RS, res.state)
    first = [g.clone() for g in be.backward(torch.empty(0), gi, *args)]
    other = be.backward(torch.empty(0), 3.0 * gi, *args)
    again = be.backward(torch.empty(0), gi, *args)
    assert any(float(g.abs().max()) > 0 for g in first)
    for a, b, c in zip(first, again, other):
        assert torch.equal(a, b)
        assert not torch.equal(a, c) or float(a.abs().max()) == 0.0
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    g = oracle.backward(f, S, gi.numpy())
    for t, k in zip(first, helpers.GRAD_KEYS):
        assert helpers.rel_inf(t.numpy().reshape(g[k].shape), g[k]) < 1e-4, k


def test_empty_and_fully_culled(sim_backend, oracle):
    params, view = make_s0(n=32)
    S, RS = helpers.settings_pair(view, bg=(0.1, 0.2, 0.3))
    empty = {k: v[:0].contiguous() for k, v in params.items()}
    res = sim_backend.forward(*[empty[k] for k in helpers.NAMES], RS)
    assert res.state[:2] == (0, 0) and torch.allclose(res.image, torch.tensor([0.1, 0.2, 0.3])[:, None, None].expand(3, 128, 128))
    grads = sim_backend.backward(torch.empty(0), torch.ones(3, 128, 128), res.image, empty['means'], empty['scales'], empty['rotations'],
                                 empty['opacities'], empty['sh_coefficients_rest'], res.buffers, RS, res.state)
    assert all(g.numel() == 0 for g in grads)
    culled = {k: v.clone() for k, v in params.items()}
    culled['means'][:, 2] = -10.0
    res = sim_backend.forward(*[culled[k] for k in helpers.NAMES], RS)
    assert res.state[:2] == (0, 0)
    grads = sim_backend.backward(torch.empty(0), torch.ones(3, 128, 128), res.image, culled['means'], culled['scales'], culled['rotations'],
                                 culled['opacities'], culled['sh_coefficients_rest'], res.buffers, RS, res.state)
    assert all(float(g.abs().max()) == 0.0 for g in grads)


@pytest.mark.parametrize('to_chw,clamp', [(True, True), (False, True), (False, False)])
def test_inference_variants(sim_backend, oracle, to_chw, clamp):
    params, view = make_s0()
    params['sh_coefficients_0'] = params['sh_coefficients_0'] * 3.0      # push colours outside [0,1] so the clamp matters
    S, RS = helpers.settings_pair(view, bg=(0.3, 0.1, 0.9))
    img = sim_backend.inference(*[params[k] for k in helpers.NAMES], RS, to_chw, clamp)
    f = oracle.forward(*helpers.np_params(params), S, inference=True, to_chw=to_chw, clamp_output=clamp)
    assert np.array_equal(img.numpy(), f['image'])


def test_adam_single_and_multi(sim_backend, oracle):
    rng = np.random.default_rng(0)
    sizes = [3000, 3000, 45001, 1000, 7, 4000]
    P = [rng.standard_normal(s).astype(np.float32) for s in sizes]
    G = [rng.standard_normal(s).astype(np.float32) for s in sizes]
    M = [np.zeros(s, np.float32) for s in sizes]; V = [np.zeros(s, np.float32) for s in sizes]
    tp, tg, tm, tv = ([torch.from_numpy(x.copy()) for x in L] for L in (P, G, M, V))
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3]
    for step in (1, 2, 3):
        for i in range(6):
            oracle.adam_step(G[i], P[i], M[i], V[i], step, lrs[i])
        sim_backend.adam_step_multi(tg, tp, tm, tv, [step] * 6, lrs, 0.9, 0.999, 1e-15)
    for i in range(6):
        assert np.array_equal(tp[i].numpy(), P[i]) and np.array_equal(tm[i].numpy(), M[i]) and np.array_equal(tv[i].numpy(), V[i])
    sim_backend.adam_step(tg[4], tp[4], tm[4], tv[4], 4, lrs[4], 0.9, 0.999, 1e-15)
    oracle.adam_step(G[4], P[4], M[4], V[4], 4, lrs[4])
    assert np.array_equal(tp[4].numpy(), P[4])


def fused_equals_backward_then_adam(sim_backend, params, view, K=16, aa=False, unaligned=False):
    """fused == backward -> FusedAdam.step() for all six groups, including invisible Gaussians, bit for bit in the simulator.
    `unaligned`: the fused side's parameters and moments start 4 bytes past a 16-byte boundary (the fused kernel's scalar path)."""
    n = params['means'].shape[0]
    S, RS = helpers.settings_pair(view, K, aa)
    order = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3]
    ref_p = {k: params[k].clone() for k in order}
    ref_m = {k: torch.randn_like(params[k]) * 1e-3 for k in order}
    ref_v = {k: torch.rand_like(params[k]) * 1e-6 for k in order}
    fus_p, fus_m, fus_v = ({k: d[k].clone() for k in order} for d in (ref_p, ref_m, ref_v))
    if unaligned:
        def odd(t):
            base = torch.empty(t.numel() + 8, dtype=t.dtype)
            shift = (1 - base.data_ptr() // 4) % 4                      # first float whose address is 4 bytes past a 16-byte boundary
            o = base[shift:shift + t.numel()].view(t.shape); o.copy_(t)
            assert o.data_ptr() % 16 == 4
            return o
        fus_p, fus_m, fus_v = ({k: odd(d[k]) for k in order} for d in (fus_p, fus_m, fus_v))
    gi = torch.randn(3, view.height, view.width, generator=torch.Generator().manual_seed(2))
    dens_ref, dens_fus = torch.zeros(2, n), torch.zeros(2, n)
    for step in (1, 2):
        res = sim_backend.forward(*[ref_p[k] for k in helpers.NAMES], RS)
        grads = sim_backend.backward(dens_ref, gi, res.image, ref_p['means'], ref_p['scales'], ref_p['rotations'], ref_p['opacities'],
                                     ref_p['sh_coefficients_rest'], res.buffers, RS, res.state)
        gmap = dict(zip(helpers.NAMES, grads))
        sim_backend.adam_step_multi([gmap[k] for k in order], [ref_p[k] for k in order], [ref_m[k] for k in order],
                                    [ref_v[k] for k in order], [step] * 6, lrs, 0.9, 0.999, 1e-15)
        res2 = sim_backend.forward(*[fus_p[k] for k in helpers.NAMES], RS)
        sim_backend.backward_adam_fused(dens_fus, gi, res2.image, [fus_p[k] for k in order], [fus_m[k] for k in order],
                                        [fus_v[k] for k in order], res2.buffers, RS, res2.state, step, lrs)
        for k in order:
            assert torch.equal(fus_p[k], ref_p[k]), (step, k)
            assert torch.equal(fus_m[k], ref_m[k]) and torch.equal(fus_v[k], ref_v[k]), (step, k)
    assert torch.equal(dens_ref, dens_fus)


@pytest.mark.parametrize('K,unaligned', [(16, False), (4, False), (16, True)])
def test_fused_backward_adam_equals_backward_then_adam(sim_backend, oracle, K, unaligned):
    params, view = make_s0(n=400)
    params['means'][:40, 2] = -10.0                      # some invisible Gaussians: zero grad, moments still decay
    fused_equals_backward_then_adam(sim_backend, params, view, K, False, unaligned)


@pytest.mark.parametrize('n,w,h,focal', [(20_000, 320, 180, 237.0), (60_001, 640, 360, 473.0)])
def test_garden_like_mid_size_scenes(sim_backend, oracle, n, w, h, focal):
    """The bench's scene generator at sizes where every kernel runs many workgroups (multi-block sort passes, several buckets per tile, tiles
    with hundreds of instances): forward intermediates and image bit-exact against the oracle, gradients to 1e-5."""
    from harness.scenes import make_garden_like, orbit_views
    params = make_garden_like(n)
    params['scales'] = params['scales'] + (1.0 if n < 50_000 else 0.7)        # footprints as large, in tiles, as the 1080p scenes' (cf. test_gpu_parity)
    res, f = _run(sim_backend, oracle, params, orbit_views(8, width=w, height=h, focal=focal)[1])
    assert f['V'] > n // 3 and f['B'] > f['ranges'].shape[0] // 2


def test_error_reporting(sim_backend):
    params, view = make_s0(n=8)
    _, RS = helpers.settings_pair(view)
    bad = RS._replace(width=0)
    with pytest.raises(RuntimeError, match='image size'):
        sim_backend.forward(*[params[k] for k in helpers.NAMES], bad)
    with pytest.raises(RuntimeError, match='contiguous float32'):
        sim_backend.forward(params['means'].double(), *[params[k] for k in helpers.NAMES[1:]], RS)


@pytest.mark.parametrize('w,h,n', [(1, 1, 40), (7, 5, 3), (16, 12, 1), (31, 23, 65)])
def test_degenerate_image_and_set_sizes(sim_backend, oracle, w, h, n):
    """Images smaller than a tile, a single pixel, one Gaussian, one more than a wavefront."""
    p, v = make_s0(seed=11, n=n)
    p['means'][:, :2] *= 0.05                       # keep them in front of the tiny image
    v = View(v.w2c, v.position, w, h, 20.0, 20.0, w / 2.0, h / 2.0, 0.2, 1e4, torch.tensor([0.3, 0.1, 0.6]))
    _run(sim_backend, oracle, p, v)


@pytest.mark.parametrize('sh_bases', [1, 4])
def test_models_with_fewer_sh_bands(sim_backend, oracle, sh_bases):
    """sh_coefficients_rest with 0 or 3 bases (SH degree 0 / 1 models): empty tensors, generic-R kernels, fused path."""
    p, v = make_s0(seed=5, n=120, sh_bases=sh_bases)
    v = View(v.w2c, v.position, 40, 30, 32.0, 32.0, 20.0, 15.0, 0.2, 1e4, torch.zeros(3))
    res, f = _run(sim_backend, oracle, p, v, K=sh_bases)
    _, RS = helpers.settings_pair(v, sh_bases)
    order = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
    P = [p[k].clone() for k in order]
    M, V = [torch.zeros_like(t) for t in P], [torch.zeros_like(t) for t in P]
    gi = torch.randn(3, 30, 40, generator=torch.Generator().manual_seed(1))
    res2 = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)
    sim_backend.backward_adam_fused(None, gi, res2.image, P, M, V, res2.buffers, RS, res2.state, 1, [1e-3] * 6)
    grads = sim_backend.backward(None, gi, res2.image, p['means'], p['scales'], p['rotations'], p['opacities'], p['sh_coefficients_rest'],
                                 res2.buffers, RS, res2.state)
    gmap = dict(zip(helpers.NAMES, grads))
    for k, t, m in zip(order, P, M):                # first Adam step: m = (1 - beta1) * g
        assert t.shape == p[k].shape
        assert torch.allclose(m, 0.1 * gmap[k], rtol=1e-4, atol=1e-9), k


def test_forward_without_host_synchronisation(sim_backend, oracle):
    """fgs_forward_async: buffers and launches sized by an instance capacity, counts read on the device. With enough capacity every result
    equals the synchronous pass; with too little the excess instances are dropped, the flag is raised, nothing is written out of bounds."""
    p, v = make_s0(seed=11, n=900)
    p['means'][:, :2] *= 0.3
    S, RS = helpers.settings_pair(v)
    args = [p[k] for k in helpers.NAMES]
    sync = sim_backend.forward(*args, RS)
    n_inst = sync.state[1]
    res = sim_backend.forward(*args, RS, instance_capacity=int(1.3 * n_inst) + 1000)
    host, _ = sim_backend.forward_counts(res, 900)
    assert host.tolist() == [sync.state[0], n_inst, 0] and res.state[:2] == (900, int(1.3 * n_inst) + 1000)
    assert torch.equal(res.image, sync.image)
    gi = torch.randn(3, v.height, v.width, generator=torch.Generator().manual_seed(0))
    back = lambda r: sim_backend.backward(torch.zeros(2, 900), gi, r.image, p['means'], p['scales'], p['rotations'], p['opacities'],
                                          p['sh_coefficients_rest'], r.buffers, RS, r.state)
    for a, b in zip(back(res), back(sync)):
        assert torch.equal(a, b)
    exact = sim_backend.forward(*args, RS, instance_capacity=n_inst)                  # exactly enough
    assert sim_backend.forward_counts(exact, 900)[0].tolist()[2] == 0 and torch.equal(exact.image, sync.image)
    small = helpers.poisoned(sim_backend).forward(*args, RS, instance_capacity=n_inst // 2)
    host, _ = sim_backend.forward_counts(small, 900)
    assert host.tolist() == [sync.state[0], n_inst, 1] and torch.isfinite(small.image).all()
    grads = back(small)                                                                # backward over the truncated lists stays in bounds
    assert all(torch.isfinite(g).all() for g in grads)


def test_equal_depth_keys_keep_every_order_independent_quantity(sim_backend, oracle):
    """Hundreds of exactly equal depth keys: counts, bounds, sorted keys, per-tile instance sets, depth order inside the lists and the final
    transmittance agree with the oracle (helpers.check_order_independent_quantities); image and gradients are order-dependent among ties."""
    p, view = helpers.tied_depth_scene()
    S, RS = helpers.settings_pair(view)
    res = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    dec = helpers.decode_forward(sim_backend, res, p['means'].shape[0], view.width, view.height)
    helpers.check_order_independent_quantities(dec, f, view.width, view.height)


def test_tile_columns_beyond_1024_use_escape_rows_sim(sim_backend, oracle):
    """The CPU half of tests/test_gpu_parity.py::test_tile_columns_beyond_1024_use_escape_rows: instance lists, ranges and image bit for bit."""
    p, v = helpers.wide_image_scene()
    S, RS = helpers.settings_pair(v)
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    sb = f['screen_bounds'].astype(np.int64)
    assert ((sb[:, 0] // 16 >= 1024) & (f['n_touched'] > 0)).sum() > 100 and ((sb[:, 0] // 16 < 1024) & (f['n_touched'] > 0)).sum() > 100
    res = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(sim_backend, res, 3000, v.width, v.height)
    assert dec['I'] == f['I'] and np.array_equal(dec['inst_keys'], f['inst_keys']) and np.array_equal(dec['inst_prims'], f['inst_prims'])
    assert np.array_equal(dec['ranges'], f['ranges']) and np.array_equal(dec['offsets'], f['offsets'])
    assert float(np.abs(res.image.numpy() - f['image']).max()) < 1e-6


def test_more_big_footprints_than_a_sort_workgroup_collects_sim(sim_backend, oracle):
    """The big-footprint list built by the depth sort's last pass, beyond its per-workgroup LDS stage: instance lists and image bit for bit."""
    p, v = helpers.many_big_footprints_scene()
    S, RS = helpers.settings_pair(v)
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    sb = f['screen_bounds'].astype(np.int64)
    n_max = ((sb[:, 1] + 15) // 16 - sb[:, 0] // 16) * ((sb[:, 3] + 11) // 12 - sb[:, 2] // 12)
    assert ((n_max > 256) & (f['n_touched'] > 0)).sum() > 300
    res = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(sim_backend, res, 700, v.width, v.height)
    assert dec['I'] == f['I'] and np.array_equal(dec['inst_keys'], f['inst_keys']) and np.array_equal(dec['inst_prims'], f['inst_prims'])
    assert np.array_equal(dec['offsets'], f['offsets']) and float(np.abs(res.image.numpy() - f['image']).max()) < 1e-6


@pytest.mark.parametrize('near,far,zscale,passes', helpers.DEPTH_RANGE_CASES)
def test_depth_sort_pass_counts_through_the_forward_sim(sim_backend, oracle, near, far, zscale, passes):
    """The depth sort's value / footprint-row handling for 4, 3, 2 and 1 passes (the single pass makes up the values AND gathers the rows): instance
    lists and image bit for bit against the oracle (the simulation compacts in index order, as the oracle does: ties included)."""
    import struct
    p, v = helpers.depth_range_scene(near, far, zscale)
    span = struct.unpack('<I', struct.pack('<f', far))[0] - struct.unpack('<I', struct.pack('<f', near))[0]
    assert (max(1, span.bit_length()) + 8) // 9 == passes
    S, RS = helpers.settings_pair(v)
    f = oracle.forward(*helpers.np_params(p), S, bucket_size=64)
    res = sim_backend.forward(*[p[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(sim_backend, res, 800, v.width, v.height)
    assert f['V'] == 800 and dec['I'] == f['I'] and np.array_equal(dec['inst_keys'], f['inst_keys']) and np.array_equal(dec['inst_prims'], f['inst_prims'])
    assert np.array_equal(dec['offsets'], f['offsets']) and float(np.abs(res.image.numpy() - f['image']).max()) < 1e-6


def test_blend_kernels_on_the_backends_own_records(sim_backend, oracle):
    """K10 / K11 isolated from K1 (helpers.check_blend_on_device_records): the oracle's blend re-run on the records the backend's K1 produced."""
    params, view = make_s0()
    r = helpers.check_blend_on_device_records(sim_backend, oracle, params, view, label='S0')
    assert r['image'] == 0.0
    p, view, K, aa, label = helpers.fuzz_configuration(8)
    helpers.check_blend_on_device_records(sim_backend, oracle, p, view, K, aa, label=label, max_masked=2e-2)


def test_records_against_fp64_conditioning_aware(sim_backend, oracle):
    """helpers.check_records_against_f64 on the simulation (bit-identical to the fp32 oracle here: the excess is the floor)."""
    params, view = make_s0()
    r = helpers.check_records_against_f64(sim_backend, oracle, params, view, label='S0')
    assert r['visible'] == 1000 and all(r[k]['differ_between_the_fp32_runs'] == 0 for k in ('mean2d', 'conic_opacity', 'color'))
