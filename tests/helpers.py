"""Shared test helpers: scene/setting conversion, decoding of the backend's private buffers, comparison metrics."""
from __future__ import annotations

import sys
import types
from pathlib import Path

import os

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / 'faster-gaussian-splatting_amd'
NAMES = ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')
GRAD_KEYS = ('means', 'scales', 'rotations', 'opacities', 'sh0', 'sh_rest')


def backend_modules():
    """FasterGSCudaBackend._lib / ._backend. The package __init__ dlopens libfgs_hip.so; if that is impossible (library not
    built yet) fall back to a bare namespace package so the ctypes glue can still be bound to the simulation library."""
    if 'FasterGSCudaBackend' not in sys.modules:
        try:
            import FasterGSCudaBackend  # noqa: F401
        except ImportError:
            pkg = types.ModuleType('FasterGSCudaBackend')
            pkg.__path__ = [str(PKG / 'FasterGSCudaBackend')]
            sys.modules['FasterGSCudaBackend'] = pkg
    from FasterGSCudaBackend import _backend, _lib
    return _lib, _backend


def sim_backend(product=False):
    """product=True: the simulation built WITHOUT -DFGS_DEV_SWITCHES, i.e. the host paths and the single formulations of libfgs_hip.so."""
    _lib, _backend = backend_modules()
    from tests.sim.build_sim import build
    return _backend.Backend(_lib.bind(build(product=product)))


def settings_pair(view, active_sh_bases=16, proper_aa=False, bg=None, device='cpu'):
    """(oracle.Settings, RasterizerSettings) for a harness View."""
    from oracle import oracle as O
    _lib, _backend = backend_modules()
    bg_t = view.background_color if bg is None else torch.tensor(bg, dtype=torch.float32)
    S = O.Settings(view.w2c.numpy(), view.position.numpy(), bg_t.numpy(), active_sh_bases, view.width, view.height, view.focal_x,
                   view.focal_y, view.center_x, view.center_y, view.near_plane, view.far_plane, proper_aa)
    RS = _backend.RasterizerSettings(view.w2c.to(device), view.position.to(device), bg_t.to(device), active_sh_bases, view.width,
                                     view.height, view.focal_x, view.focal_y, view.center_x, view.center_y, view.near_plane,
                                     view.far_plane, proper_aa)
    return S, RS


def np_params(params):
    return [params[k].detach().cpu().numpy() for k in NAMES]


def decode_forward(be, res, n, width, height):
    """Pulls every intermediate out of the backend's private buffers (layout from fgs_blob_layout) as numpy arrays."""
    nv, ni, nb, sel = res.state
    out = {'V': nv, 'I': ni, 'B_cap': nb}
    bufs = [b.cpu() for b in res.buffers]
    lp = be.blob_layout(0, n, width, height, ni, nb)
    rec = be.view(bufs[0], lp, 'rec', torch.float32).reshape(n, 12).numpy()
    recu = rec.view(np.uint32)
    out['mean2d'], out['conic_opacity'] = rec[:, 0:2], rec[:, 2:6]
    out['color'] = rec[:, 6:9]
    bx, by = recu[:, 9], recu[:, 10]
    out['screen_bounds'] = np.stack([bx & 0xffff, bx >> 16, by & 0xffff, by >> 16], 1).astype(np.uint16)
    out['rec_hit_mask'] = recu[:, 11]
    out['n_touched'] = be.view(bufs[0], lp, 'n_touched', torch.int32).numpy().view(np.uint32)
    out['offsets'] = be.view(bufs[0], lp, 'offsets', torch.int32).numpy().view(np.uint32)[:nv]
    for s in (0, 1):
        out[f'depth_keys{s}'] = be.view(bufs[0], lp, f'depth_keys{s}', torch.int32).numpy().view(np.uint32)[:nv]
        out[f'prim_idx{s}'] = be.view(bufs[0], lp, f'prim_idx{s}', torch.int32).numpy().view(np.uint32)[:nv]
    lt = be.blob_layout(1, n, width, height, ni, nb)
    out['ranges'] = be.view(bufs[1], lt, 'ranges', torch.int32).reshape(-1, 2).numpy().view(np.uint32)
    if 'tile_plan' in lt and bufs[1].numel() >= lt['tile_plan'][0] + lt['tile_plan'][1]:
        out['tile_plan'] = be.view(bufs[1], lt, 'tile_plan', torch.int32).numpy().view(np.uint32)
        out['bucket_offsets_any_mode'] = be.view(bufs[1], lt, 'bucket_offsets', torch.int32).numpy().view(np.uint32)
    if 'bucket_offsets' in lt and bufs[1].numel() >= lt['n_processed'][0] + lt['n_processed'][1]:
        out['bucket_offsets'] = be.view(bufs[1], lt, 'bucket_offsets', torch.int32).numpy().view(np.uint32)
        out['max_n_processed'] = be.view(bufs[1], lt, 'max_n_processed', torch.int32).numpy().view(np.uint32)
        out['final_T_tiles'] = be.view(bufs[1], lt, 'final_T', torch.float32).reshape(-1, 192).numpy()
        out['n_processed_tiles'] = be.view(bufs[1], lt, 'n_processed', torch.int32).reshape(-1, 192).numpy().view(np.uint32)
    li = be.blob_layout(2, n, width, height, ni, nb)
    gw, gh = (width + 15) // 16, (height + 11) // 12
    key_dtype = torch.int16 if (gw * gh - 1).bit_length() <= 16 else torch.int32
    keys = be.view(bufs[2], li, f'keys{sel}', key_dtype).numpy()[:ni]
    out['inst_keys'] = keys.view(np.uint16 if key_dtype == torch.int16 else np.uint32).astype(np.uint32)
    out['inst_prims'] = be.view(bufs[2], li, f'prims{sel}', torch.int32).numpy().view(np.uint32)[:ni]
    if len(bufs) > 3 and nb > 0 and 'bucket_offsets' in out:
        lb = be.blob_layout(3, n, width, height, ni, nb)
        B = int(out['bucket_offsets'][-1])
        out['B'] = B
        out['bucket_tile_index'] = be.view(bufs[3], lb, 'tile_index', torch.int32).numpy().view(np.uint32)[:B]
        out['bucket_ckpt'] = be.view(bufs[3], lb, 'ckpt', torch.float32).reshape(-1, 192, 4).numpy()[:B]
    return out


def tiles_to_image(tile_major: np.ndarray, width: int, height: int, fill=0):
    """[T,192] tile-major (row-major inside a 16x12 tile) -> [H,W]."""
    gw, gh = (width + 15) // 16, (height + 11) // 12
    full = tile_major.reshape(gh, gw, 12, 16).transpose(0, 2, 1, 3).reshape(gh * 12, gw * 16)
    return full[:height, :width]


def _log_tolerance(kind: str, value: float, a: np.ndarray, ref: np.ndarray, **extra) -> None:
    """FGS_TOL_LOG=<file>: one line per metric evaluation (call site, achieved value, and what the SAME data gives under the
    1e-4-of-max-abs bar) -- how much slack every tolerance in the suite really has. Off by default; changes no result."""
    path = os.environ.get('FGS_TOL_LOG')
    if not path or not ref.size:
        return
    import inspect
    fr = inspect.stack()[2]
    site = next((f for f in inspect.stack()[2:] if os.path.basename(f.filename).startswith('test_')), fr)
    err, scale = np.abs(a - ref), np.abs(ref).max() + 1e-30
    with open(path, 'a') as fh:
        fh.write(f'{os.path.basename(site.filename)}:{site.lineno} {site.function} {kind}={value:.3e} rel_inf={err.max() / scale:.3e} '
                 f'frac_above_1e-4_of_max={float((err > 1e-4 * scale).mean()):.3e} n={ref.size} {extra}\n')


def log_note(kind: str, value, **extra) -> None:
    """FGS_TOL_LOG: a counted event that is not a tensor comparison (e.g. how often the integer-mismatch budget of a forward check is used)."""
    path = os.environ.get('FGS_TOL_LOG')
    if not path:
        return
    import inspect
    site = next((f for f in inspect.stack()[1:] if os.path.basename(f.filename).startswith('test_')), inspect.stack()[1])
    with open(path, 'a') as fh:
        fh.write(f'{os.path.basename(site.filename)}:{site.lineno} {site.function} {kind}={value} {extra}\n')


def rel_inf(a: np.ndarray, ref: np.ndarray) -> float:
    """max |a-ref| normalised by max |ref| -- the tolerance metric of the float parity tests."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    value = float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30)) if ref.size else 0.0
    _log_tolerance('rel_inf', value, a, ref)
    return value


ELEM_RTOL = 1e-4            # north_star: "within 1e-4 relative float" -- per ELEMENT, not per tensor
ELEM_FRACTION = 1e-4        # entries allowed beyond it (fp32 sums that cancel to << their terms carry the error of the terms)


def elementwise_fraction(a: np.ndarray, ref: np.ndarray, keep: np.ndarray | None = None, rtol: float = ELEM_RTOL, kind: str = 'elementwise',
                         free_rows: int = 0) -> float:
    """Second criterion beside rel_inf (missing #3): the fraction of entries with |a - ref| > rtol * |ref| + atol, where
    atol = rtol * MEDIAN |ref| over the non-zero reference entries -- not the tensor's maximum, so an entry a thousand times smaller
    than the largest one is still held to its own magnitude (down to the median's scale; below that fp32 accumulation order decides).
    `keep`: boolean mask over the first axis (rows outside the oracle's threshold-risk masks). `free_rows`: rows (Gaussians) that may
    fail without being counted -- for the small scenes compared WITHOUT a risk mask, where one alpha-threshold flip moves one or two
    Gaussians' gradients and 1e-4 of a few hundred entries is less than one entry. Logged like rel_inf under FGS_TOL_LOG."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    if keep is not None:
        a, ref = a[keep], ref[keep]
    if not ref.size:
        return 0.0
    mag = np.abs(ref)
    nz = mag[mag > 0]
    atol = rtol * float(np.median(nz)) if nz.size else 0.0
    excess = np.abs(a - ref) / (rtol * mag + atol + 1e-300)
    bad = excess > 1.0
    n_bad = int(bad.sum())
    if free_rows and n_bad:
        rows = bad.reshape(bad.shape[0], -1).any(axis=1)
        if int(rows.sum()) <= free_rows:
            n_bad = 0
    value = n_bad / ref.size
    _log_tolerance(kind, value, a, ref, rtol=rtol, atol=atol, worst_excess=float(excess.max()), bad_entries=int(bad.sum()))
    return value


# The HIP path may miss the element-wise bar (against the fp64 evaluation) at most this often relative to the fp32 oracle. Measured on MI355X:
# 0.98-1.03 at every site with >= 60 k Gaussians (profiles/archive/r03_gpu_tolerance_slack.txt), so those are held to 1.10 (round 4); the small scenes, where the
# counts are a handful of entries, keep 1.25 beside their 4-sigma term.
THREE_WAY_FACTOR = 1.25
THREE_WAY_FACTOR_LARGE = 1.10
LARGE_SCENE_ROWS = 60_000


def elementwise_three_way(a, ref32, truth, keep: np.ndarray | None = None, kind: str = '', free_rows: int = 0):
    """The element-wise 1e-4 bar cannot be held by ANY fp32 evaluation of these gradients against another one: they are sums of
    hundreds of signed per-pixel terms, and an entry that cancels to 1 % of its terms carries 100x the relative rounding error.
    Measured (tests/test_oracle.py::test_fp32_oracle_misses_elementwise_bar_by_conditioning): on a deep 1500-Gaussian scene the fp32
    ORACLE misses the bar against the same formulas evaluated in double (oracle.forward_backward_f64, itself within 6e-8 of the
    independent fp64 autograd model) on 1.3-3.9 % of the entries of four of the six gradient tensors. So the bar is applied three-way:
    both fp32 results are measured against the fp64 values, and the HIP path may miss it at most THREE_WAY_FACTOR times as often as the
    reference-arithmetic oracle does, plus the ELEM_FRACTION budget. Returns (fraction HIP vs fp64, fraction oracle32 vs fp64, entries compared); the
    direct HIP-vs-oracle32 fraction (the two fp32 noises added) is logged beside them."""
    fh = elementwise_fraction(a, truth, keep, kind='elem_hip_vs_f64_' + kind, free_rows=free_rows)
    fo = elementwise_fraction(ref32, truth, keep, kind='elem_oracle32_vs_f64_' + kind)
    elementwise_fraction(a, ref32, keep, kind='elem_hip_vs_oracle32_' + kind)          # logged only
    n = int(np.asarray(truth)[keep].size if keep is not None else np.asarray(truth).size)
    return fh, fo, n


def three_way_ok(frac_hip: float, frac_oracle: float, n: int, cluster: int = 1) -> bool:
    # n / cluster = rows (Gaussians, or pixels for an image) of the compared tensor: LARGE_SCENE_ROWS and more -> the tighter factor
    """frac_hip <= THREE_WAY_FACTOR * frac_oracle + ELEM_FRACTION, plus four standard deviations of a count of n * frac_oracle entries
    (the two fp32 roundings are independent: on a 900-entry tensor 'the oracle misses 3, the HIP path 5' is noise, not a finding). Four, not
    three: the suite applies this to ~1500 tensors and a wide fuzz sweep to 1200 more, and a 3-sigma bar fires once in 740 on pure noise --
    it did (round 3, seed 115 of the sweep: 5 entries of 3000 against 1, both max-norm errors below 1e-6)."""
    # `cluster`: entries that miss the bar together -- the 3 / 4 / 45 gradient entries of ONE Gaussian share their ill-conditioned sums, so
    # the count fluctuates like n / cluster independent events of `cluster` entries each (seed 411 of the sweep: 10 entries = 3-4 Gaussians
    # of 480 against 2 entries = 1 Gaussian, every max-norm error below 4e-6)
    sigma = (cluster * max(frac_oracle, float(cluster) / max(n, 1)) / max(n, 1)) ** 0.5
    factor = THREE_WAY_FACTOR_LARGE if n // max(cluster, 1) >= LARGE_SCENE_ROWS else THREE_WAY_FACTOR
    log_note('three_way_ratio', f'{(frac_hip / frac_oracle if frac_oracle > 0 else 0.0):.3f}', factor=factor, rows=n // max(cluster, 1), frac_hip=f'{frac_hip:.3e}', frac_oracle=f'{frac_oracle:.3e}')
    return frac_hip <= factor * frac_oracle + ELEM_FRACTION + 4.0 * sigma


def outlier_fraction(a: np.ndarray, ref: np.ndarray, rtol: float, atol: float) -> float:
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    bad = np.abs(a - ref) > (atol + rtol * np.abs(ref))
    value = float(bad.mean()) if ref.size else 0.0
    _log_tolerance('outlier_fraction', value, a, ref, rtol=rtol, atol=atol)
    return value


def check_forward_against_oracle(dec: dict, f: dict, exact_floats: bool, width: int, height: int, image: np.ndarray,
                                 int_mismatch_budget: int = 0, pixel_mask: np.ndarray | None = None, max_masked_pixels: float = 1e-3):
    """Integer intermediates bit-exact (up to `int_mismatch_budget` primitives whose libm-ULP-sensitive bounds differ);
    float intermediates within 1e-5 relative (exact when both sides use the same libm, i.e. the simulation).
    On hardware (`exact_floats` False) the per-pixel outputs are held to 1e-4 outside `pixel_mask` -- flip_masks()['pixel'], the pixels
    that own a (pixel, Gaussian) pair within an ULP-scale margin of the alpha / transmittance thresholds -- whose size is bounded (1e-3)."""
    vis = f['n_touched'] > 0
    assert dec['V'] == f['V'] and dec['I'] == f['I'], (dec['V'], f['V'], dec['I'], f['I'])
    nt_bad = int((dec['n_touched'] != f['n_touched']).sum())
    sb_bad = int((dec['screen_bounds'][vis] != f['screen_bounds'][vis]).any(axis=1).sum())
    assert nt_bad <= int_mismatch_budget and sb_bad <= int_mismatch_budget, (nt_bad, sb_bad)
    for k in ('mean2d', 'conic_opacity', 'color'):
        if exact_floats:
            assert np.array_equal(dec[k][vis], f[k][vis]), k
        else:
            assert rel_inf(dec[k][vis], f[k][vis]) < 1e-5, (k, rel_inf(dec[k][vis], f[k][vis]))
    if int_mismatch_budget == 0:
        sel = 0 if np.array_equal(dec['prim_idx0'], f['prim_idx']) else 1
        assert np.array_equal(dec[f'prim_idx{sel}'], f['prim_idx']) and np.array_equal(dec[f'depth_keys{sel}'], f['depth_keys'])
        assert np.array_equal(dec['offsets'], f['offsets'])
        assert np.array_equal(dec['inst_keys'], f['inst_keys']) and np.array_equal(dec['inst_prims'], f['inst_prims'])
        assert np.array_equal(dec['ranges'], f['ranges'])
        if 'bucket_offsets' in dec:
            assert np.array_equal(dec['bucket_offsets'], f['bucket_offsets'])
            if 'bucket_tile_index' in dec:                          # absent when the pass produced no bucket (nothing visible)
                assert np.array_equal(dec['bucket_tile_index'], f['bucket_tile_index'])
            else:
                assert f['B'] == 0
    if 'n_processed_tiles' in dec:
        npr = tiles_to_image(dec['n_processed_tiles'], width, height)
        fT = tiles_to_image(dec['final_T_tiles'], width, height)
        if exact_floats:
            assert np.array_equal(npr.reshape(-1), f['n_processed']) and np.array_equal(dec['max_n_processed'], f['max_n_processed'])
            assert np.array_equal(fT.reshape(-1), f['final_T'])
        else:
            assert pixel_mask is not None, 'hardware comparison needs the threshold-risk mask of the oracle (flip_masks)'
            keep = ~pixel_mask.reshape(-1)
            assert float(pixel_mask.mean()) < max_masked_pixels, ('masked pixels', float(pixel_mask.mean()))
            assert np.array_equal(npr.reshape(-1)[keep], f['n_processed'][keep]), int((npr.reshape(-1)[keep] != f['n_processed'][keep]).sum())
            assert np.abs(fT.reshape(-1)[keep] - f['final_T'][keep]).max() < 1e-4, float(np.abs(fT.reshape(-1)[keep] - f['final_T'][keep]).max())
    if exact_floats:
        assert np.array_equal(image, f['image'])
    else:
        assert pixel_mask is not None, 'hardware comparison needs the threshold-risk mask of the oracle (flip_masks)'
        err = np.abs(np.asarray(image, np.float64) - f['image']).max(axis=0)
        scale = max(1.0, float(np.abs(f['image']).max()))
        assert float(pixel_mask.mean()) < max_masked_pixels, ('masked pixels', float(pixel_mask.mean()))
        assert err[~pixel_mask].max() < 1e-4 * scale, ('image outside the mask', float(err[~pixel_mask].max()))
        assert err.max() < 5e-3, ('image inside the mask', float(err.max()))


def poisoned(be):
    """A Backend whose scratch buffers arrive filled with 0xFF bytes (= NaN floats / huge integers): nothing the kernels
    read may depend on the previous contents of freshly (re)sized scratch memory."""
    _lib, _backend = backend_modules()

    class Poisoned(_backend.Backend):
        @staticmethod
        def _make_resizer(device, n_buffers):
            buffers = [torch.empty(0, dtype=torch.uint8, device=device) for _ in range(n_buffers)]

            def resize(_user, which, nbytes):
                buffers[which].resize_(int(nbytes))
                buffers[which].fill_(255)
                return buffers[which].data_ptr() if nbytes else 0
            return buffers, _lib.RESIZE_FN(resize)

        def _scratch(self, n, settings, device):
            t = super()._scratch(n, settings, device)
            t.fill_(255)
            return t
    return Poisoned(be.lib)


# ---- flip-aware parity ("what's weak" 3): count and exclude exactly the entries that sit on a hard threshold ------
def flip_masks(oracle, f, S, dec=None, eps=5e-6, eps_T=1e-5):
    """Masks of the outputs that an ULP-level difference in exp / FMA contraction can legitimately move by more than 1e-4:
    pixels and Gaussians with a (pixel, Gaussian) pair within `eps` (relative) of the alpha >= 1/255 test or a transmittance within
    `eps_T` of the termination test (oracle.threshold_risk), plus -- if the decoded HIP intermediates are given -- Gaussians whose integer
    screen bounds / tile count differ (a floor / ceil / cull input within an ULP of its threshold in preprocess).
    eps: v_exp_f32 (1 ulp) after the x * log2(e) rounding at |x| <= 5.6, plus FMA contraction of the three-term exponent, move alpha
    by <= ~2e-6 relative against glibc expf without contraction; 5e-6 leaves a margin of 2.5. Measured on MI355X (tools/archive/diag_flip.py,
    S1 / S2 at 1080p): every pixel that differs by more than 1e-4 lies inside this mask, and outside it the image agrees to 5e-5 and
    all six gradients to 3e-5 of their max-abs value."""
    r = oracle.threshold_risk(f, S, eps, eps_T)
    if dec is not None and dec.get('I') == f['I'] and 'conic_opacity' in dec and np.array_equal(dec['screen_bounds'], f['screen_bounds']):
        # ... and the same walk over the DEVICE's records (round 6). The two sets of records are not bit-identical everywhere: a needle-shaped Gaussian's
        # conic is cov / det with det a cancelling difference, and the one-ulp difference between the device's exp and glibc's in the variances reaches
        # 7e-6 of the conic (seed 10436 of a 4000-seed sweep) -- a pair 1.5e-5 above the cut by the oracle's records can sit below it by the device's.
        # A pair within the band by EITHER set of records is a legitimate flip.
        vis = f['n_touched'] > 0
        f_dev = dict(f)
        f_dev['mean2d'] = np.ascontiguousarray(np.where(vis[:, None], dec['mean2d'], f['mean2d']).astype(np.float32))
        f_dev['conic_opacity'] = np.ascontiguousarray(np.where(vis[:, None], dec['conic_opacity'], f['conic_opacity']).astype(np.float32))
        r2 = oracle.threshold_risk(f_dev, S, eps, eps_T)
        r = {k: r[k] | r2[k] for k in ('pixel', 'prim', 'near')}
    prim = r['prim'].copy()
    pixel, near = r['pixel'], r['near']
    if dec is not None:
        vis = (f['n_touched'] > 0) | (dec['n_touched'] > 0)
        prim |= vis & ((dec['n_touched'] != f['n_touched']) | (dec['screen_bounds'] != f['screen_bounds']).any(axis=1))
    return {'pixel': pixel, 'prim': prim, 'near': near & ~prim}


def masked_rel_inf(a, ref, keep):
    """rel_inf over the rows selected by the boolean `keep` (first axis), normalised by max |ref| over ALL rows."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    if not keep.any():
        return 0.0
    return float(np.abs(a[keep] - ref[keep]).max() / (np.abs(ref).max() + 1e-30))


def seeded_moments(shape, seed: int):
    """Non-zero Adam moments (exp_avg ~ 1e-3, exp_avg_sq ~ 1e-6) for tests that compare two paths through several optimizer steps: from
    zero moments the first steps are lr * g / |g|, and an entry whose tiny gradient changes sign under another summation order moves by
    2 lr -- the comparison would measure the optimizer's sensitivity, not the agreement of the gradients."""
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=gen) * 1e-3, torch.rand(shape, generator=gen) * 1e-6 + 1e-7


def seed_trainer_moments(trainer, names, seed: int = 11) -> None:
    """Writes seeded_moments() into a harness trainer's exp_avg / exp_avg_sq arenas (trainer.layout[name] = (offset, numel, shape))."""
    for i, k in enumerate(names):
        o, n, shape = trainer.layout[k]
        m0, v0 = seeded_moments(shape, seed + i)
        trainer.exp_avg[o:o + n].view(shape).copy_(m0)
        trainer.exp_avg_sq[o:o + n].view(shape).copy_(v0)


def check_flip_aware(image, f_image, grads: dict, g_ref: dict, masks: dict, tol=1e-4, max_masked=1e-3, loose=5e-2, label='', elem_fraction=ELEM_FRACTION,
                     truth: dict | None = None, near_tol: float | None = None, image_flip_budget: int = 0):
    """image [3,H,W]; grads / g_ref: {name: array with the Gaussian index first}. Entries outside the masks must agree to `tol`
    (max-abs error relative to the tensor's max-abs value) AND element by element: with `truth` (oracle.forward_backward_f64: the fp64
    values of 'image' and the six gradients) three-way (elementwise_three_way), without it directly against the oracle (fewer than
    `elem_fraction` of the entries beyond 1e-4 of their own magnitude + 1e-4 of the tensor's median magnitude); the masked fraction is
    bounded; masked entries stay within `loose`. `near_tol` (the adversarial fuzz scenes only): Gaussians that merely CONTRIBUTE to a pixel
    with a borderline pair (masks['near']) form a third class held to that looser bar -- a flipped pair in front of them scales their
    share of that pixel by 1 - 1/255, which is 2e-4 of a tensor's maximum when the pixel dominates a small Gaussian's gradient (seed 243 of
    a wide sweep, round 3); by default they are held to `tol` like everybody else."""
    report = {}
    pm = masks['pixel']
    frac_p, frac_g = float(pm.mean()), float(masks['prim'].mean()) if masks['prim'].size else 0.0
    report['masked_pixels'], report['masked_gaussians'] = frac_p, frac_g
    log_note('masked_fraction', f'{max(frac_p, frac_g):.3e}', label=label.replace(' ', '_'), pixels=f'{frac_p:.3e}', gaussians=f'{frac_g:.3e}', bound=max_masked,
             near=f'{float(masks["near"].mean()) if "near" in masks and masks["near"].size else 0.0:.3e}')
    assert frac_p < max_masked and frac_g < max_masked, (label, 'masked fraction', frac_p, frac_g)
    if image is not None:
        err = np.abs(np.asarray(image, np.float64) - f_image).max(axis=0)
        scale = max(1.0, float(np.abs(f_image).max()))
        outside = np.sort(err[~pm].reshape(-1))[::-1] / scale if (~pm).any() else np.zeros(1)
        # `image_flip_budget` pixels OUTSIDE the mask may hold an alpha-test flip the oracle's band did not name (each bounded by `loose`): only the
        # 8K test uses it -- at pixel coordinates of several thousand the exponent's terms carry ~1e-6 of rounding, more than the 5e-6 relative band
        report['image_flips_outside_mask'] = int((outside[:image_flip_budget + 1] >= tol).sum()) if image_flip_budget else 0
        assert image_flip_budget == 0 or float(outside[0]) < loose, (label, 'image (unmasked flip)', float(outside[0]))
        report['image'] = float(outside[min(image_flip_budget, outside.size - 1)])
        report['image_masked'] = float(err[pm].max() / scale) if pm.any() else 0.0
        hwc = lambda x: np.moveaxis(np.asarray(x), 0, -1)[~pm]
        assert report['image'] < tol, (label, 'image', report)
        if truth is not None and 'image' in truth:
            report['image_elem'] = elementwise_three_way(hwc(image), hwc(f_image), hwc(truth['image']), kind='image')
            assert three_way_ok(*report['image_elem'], cluster=3), (label, 'image (element-wise 1e-4, three-way)', report)
        else:
            report['image_elem'] = elementwise_fraction(hwc(image), hwc(f_image), kind='elementwise_image')
            assert report['image_elem'] < elem_fraction, (label, 'image (element-wise 1e-4)', report)
        assert report['image_masked'] < loose, (label, 'image (masked pixels)', report)
    keep = ~masks['prim']
    if near_tol is not None and 'near' in masks:
        near = masks['near'] & keep
        keep = keep & ~near
        report['near_gaussians'] = float(near.mean()) if near.size else 0.0
        # the looser class is capped: measured on hardware at most 0.13 of the Gaussians of an adversarial fuzz scene (profiles/r04_gpu_tolerance_slack.txt
        # part 4, seed 8) and 0.10 at the layered S2 scene; a class that swallows a quarter of a scene would make the 1e-4 bar meaningless
        assert report['near_gaussians'] < 0.25, (label, 'share of Gaussians in the near class', report['near_gaussians'])
        for k, a in grads.items():
            report[k + '_near'] = masked_rel_inf(np.asarray(a).reshape(g_ref[k].shape), g_ref[k], near)
            assert report[k + '_near'] < near_tol, (label, k + ' (Gaussians behind a borderline pair)', report)
    for k, a in grads.items():
        ref = g_ref[k]
        a = np.asarray(a).reshape(ref.shape)
        report[k] = masked_rel_inf(a, ref, keep)
        report[k + '_masked'] = masked_rel_inf(a, ref, masks['prim'])
        if report[k] >= tol and truth is not None and k in truth:
            # A tensor whose LARGEST entry is itself an ill-conditioned sum (a one-Gaussian scene: its opacity gradient is a signed sum over the whole
            # footprint; seeds 5645 / 5916 of a 4000-seed sweep, round 6) has no well-conditioned entry to set the scale: there the max-norm bar is
            # applied three-way like the element-wise one -- the HIP value may be at most twice as far from the fp64 value as the fp32 oracle is.
            t64 = np.asarray(truth[k]).reshape(ref.shape)
            err_hip, err_o32 = masked_rel_inf(a, t64, keep), masked_rel_inf(ref, t64, keep)
            report[k + '_vs_f64'] = (err_hip, err_o32)
            log_note('max_norm_three_way', f'{err_hip:.3e}', label=label.replace(' ', '_'), tensor=k, oracle32=f'{err_o32:.3e}', hip_vs_oracle32=f'{report[k]:.3e}')
            assert err_hip <= max(tol, 2.0 * err_o32), (label, k + ' (max-norm, three-way)', report)
        else:
            assert report[k] < tol, (label, k, report)
        assert report[k + '_masked'] < loose, (label, k + ' (masked Gaussians)', report)
        if truth is not None and k in truth:
            report[k + '_elem'] = elementwise_three_way(a, ref, np.asarray(truth[k]).reshape(ref.shape), keep, kind=k)
            assert three_way_ok(*report[k + '_elem'], cluster=int(np.prod(ref.shape[1:])) if ref.ndim > 1 else 1), (label, k + ' (element-wise 1e-4, three-way)', report)
        else:
            report[k + '_elem'] = elementwise_fraction(a, ref, keep, kind='elementwise_' + k)
            assert report[k + '_elem'] < elem_fraction, (label, k + ' (element-wise 1e-4)', report)
    return report


# ---- K10's device-side tile plan (binning.hip: plan_tiles_kernel), restated in numpy for the tests ---------------------------------------------
PLAN_BX, PLAN_BY, PLAN_HEADER = 8, 10, 4


def planned_tile_of_workgroup(plan: np.ndarray, grid_w: int, grid_h: int) -> np.ndarray:
    """tile index blended by every workgroup of the planned K10 grid (-1 = padding workgroup), from the plan words the device wrote --
    the same arithmetic as blend_forward.hip: tile_of_workgroup(row_group == kPlannedBlocks)."""
    bw, bh = -(-grid_w // PLAN_BX), -(-grid_h // PLAN_BY)
    assert (int(plan[0]), int(plan[1]), int(plan[2]), int(plan[3])) == (bw, bh, bw * bh, PLAN_BX * PLAN_BY // 8)
    per_block, per_xcd = bw * bh, PLAN_BX * PLAN_BY // 8
    wg = np.arange(PLAN_BX * PLAN_BY * per_block)
    xcd, q = wg % 8, wg // 8
    slot, local = q // per_block, q % per_block
    b = plan[PLAN_HEADER + xcd * per_xcd + slot].astype(np.int64)
    tx, ty = (b % PLAN_BX) * bw + local % bw, (b // PLAN_BX) * bh + local // bw
    return np.where((tx < grid_w) & (ty < grid_h), ty * grid_w + tx, -1)


def check_tile_plan(plan: np.ndarray, bucket_offsets: np.ndarray, grid_w: int, grid_h: int) -> dict:
    """The plan must send every tile to exactly one workgroup; every XCD gets the same number of blocks; the heaviest XCD carries at most
    the average weight plus one block (what the greedy deal guarantees); every XCD walks its blocks from heavy to light."""
    tiles = planned_tile_of_workgroup(plan, grid_w, grid_h)
    real = tiles[tiles >= 0]
    assert np.array_equal(np.sort(real), np.arange(grid_w * grid_h)), 'every tile exactly once'
    blocks = plan[PLAN_HEADER:PLAN_HEADER + PLAN_BX * PLAN_BY].astype(np.int64)
    assert np.array_equal(np.sort(blocks), np.arange(PLAN_BX * PLAN_BY)), 'every block exactly once'
    bw, bh = -(-grid_w // PLAN_BX), -(-grid_h // PLAN_BY)
    nb = np.diff(np.concatenate([[0], bucket_offsets[:grid_w * grid_h].astype(np.int64)])).reshape(grid_h, grid_w)
    weight = np.zeros(PLAN_BX * PLAN_BY, np.int64)
    for b in range(PLAN_BX * PLAN_BY):
        sub = nb[(b // PLAN_BX) * bh:(b // PLAN_BX + 1) * bh, (b % PLAN_BX) * bw:(b % PLAN_BX + 1) * bw]
        weight[b] = sub.sum() + sub.size
    per_xcd = blocks.reshape(8, -1)
    loads = weight[per_xcd].sum(axis=1)
    assert loads.max() <= loads.mean() + weight.max(), (loads, weight.max())
    assert all(np.all(np.diff(weight[row]) <= 0) for row in per_xcd), 'heaviest block first inside an XCD'
    return {'loads': loads, 'weights': weight}


def fuzz_configuration(seed: int):
    """Seeded random scene / camera / SH degree / antialiasing mode of the fuzz tests (tests/test_gpu_fuzz.py on hardware, tests/test_sim_fuzz.py in the
    CPU simulation): see the docstring of test_gpu_fuzz.py for what is being varied."""
    from harness.scenes import View, make_s0
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 500, 1000, 2047, 2049, 3000]))
    W, H = int(rng.integers(17, 420)), int(rng.integers(13, 300))
    near, far = float(rng.choice([0.01, 0.2, 1.0, 3.2])), float(rng.choice([4.6, 100.0, 1.0e4]))
    K, aa = int(rng.choice([1, 4, 9, 16])), bool(rng.integers(0, 2))
    p, v = make_s0(seed=100 + seed, n=n)
    g = torch.Generator().manual_seed(seed)
    pick = lambda frac: torch.rand(n, generator=g) < frac
    p['scales'][pick(0.03)] += 2.5                               # screen-filling: medium / huge / hot footprint paths
    p['scales'][pick(0.05)] -= 3.0                               # sub-pixel
    p['means'][pick(0.05), 2] = -9.0                             # behind the camera
    p['means'][pick(0.03), 2] = 2.0e4                            # beyond every far plane
    p['rotations'][pick(0.02)] = 0.0                             # |q|^2 < 1e-8
    p['opacities'][pick(0.05)] = float(np.log((1 / 255) / (1 - 1 / 255))) + 1e-3     # sigmoid just above the cut
    p['opacities'][pick(0.02)] = -20.0
    # Distinct view depths. Gaussians whose depth keys are bit-identical are blended in the order the preprocess kernel's counter atomics
    # arrive (the reference: one atomicAdd per Gaussian, kf:204-208; the oracle: primitive order), and swapping two such neighbours moves
    # the colour by T a_A a_B (c_B - c_A) wherever both contribute -- 1e-4 of the image in seed 126 of a wide sweep (round 3). The camera
    # of these scenes looks down z from 4 units away, so depth = z + 4 drops low bits and ~1 pair in 1000 collides: nudged apart here.
    w2c = v.w2c.numpy().astype(np.float32)
    for _ in range(8):
        m = p['means'].numpy()
        depth = ((m[:, 0] * w2c[2, 0] + m[:, 1] * w2c[2, 1]) + (m[:, 2] * w2c[2, 2] + w2c[2, 3])).astype(np.float32)
        _, first, counts = np.unique(depth.view(np.uint32), return_index=True, return_counts=True)
        if (counts > 1).sum() == 0:
            break
        dup = np.setdiff1d(np.arange(n), first)
        dup = dup[np.abs(m[dup, 2]) < 100.0]                      # not the ones parked behind the camera / beyond the far plane (culled anyway)
        if len(dup) == 0:
            break
        p['means'][torch.from_numpy(dup), 2] += 1e-4 * (1.0 + torch.arange(len(dup), dtype=torch.float32))
    focal = float(W) * float(rng.uniform(0.6, 1.4))
    bg = torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)
    view = View(v.w2c, v.position, W, H, focal, focal * float(rng.uniform(0.9, 1.1)), W / 2 + float(rng.uniform(-9, 9)), H / 2 + float(rng.uniform(-9, 9)),
                near, far, bg)
    return p, view, K, aa, f'seed {seed}: n={n} {W}x{H} near={near} far={far} K={K} aa={aa}'


# ---- equal depth keys (round-3 advisor finding: the fuzz scenes nudge depth ties apart, which hides the one legitimately order-dependent part) ----
def tied_depth_scene(n: int = 600, seed: int = 21):
    """S0 geometry with every Gaussian on one of three depth planes: the camera looks along +z, so depth = z + 4 is exactly representable and
    hundreds of depth keys are EQUAL. Among equal keys the blending order is the visible list's order -- K1's atomic compaction order on hardware
    (as in the reference, kf:204-208), the index order in the oracle -- so image and gradients may differ by more than rounding; everything else
    may not."""
    from harness.scenes import make_s0
    p, view = make_s0(seed=seed, n=n)
    g = torch.Generator().manual_seed(seed)
    p['means'][:, 2] = torch.tensor([-0.5, 0.0, 0.5])[torch.randint(0, 3, (n,), generator=g)]
    p['opacities'] = p['opacities'] - 1.5                      # no pixel terminates early: the final transmittance is a plain product
    return p, view


def check_order_independent_quantities(dec: dict, f: dict, width: int, height: int) -> None:
    """What must agree with the oracle whatever order equal depth keys end up in: counts, bounds, tile counts, the sorted depth keys, every
    tile's range and its SET of instances, non-decreasing depth inside every tile list, and (without early termination) the final transmittance."""
    assert dec['V'] == f['V'] and dec['I'] == f['I']
    assert np.array_equal(dec['n_touched'], f['n_touched'])
    vis = f['n_touched'] > 0
    assert np.array_equal(dec['screen_bounds'][vis], f['screen_bounds'][vis])
    sel = 0 if np.array_equal(np.sort(dec['prim_idx0']), np.sort(f['prim_idx'])) and np.all(np.diff(dec['depth_keys0'].astype(np.int64)) >= 0) else 1
    assert np.array_equal(dec[f'depth_keys{sel}'], f['depth_keys'])                        # the sorted keys are the same multiset in the same order
    assert np.array_equal(np.sort(dec[f'prim_idx{sel}']), np.sort(f['prim_idx']))
    assert len(np.unique(f['depth_keys'])) < f['V'] // 50                                 # ... and heavily tied
    assert np.array_equal(dec['ranges'], f['ranges'])
    assert np.array_equal(dec['inst_keys'], f['inst_keys'])                                # tile keys after the tile sort: identical (same counts per tile)
    key_of = dict(zip(f['prim_idx'].tolist(), f['depth_keys'].tolist()))
    for t, (a, b) in enumerate(f['ranges']):
        mine, ref = dec['inst_prims'][a:b], f['inst_prims'][a:b]
        assert np.array_equal(np.sort(mine), np.sort(ref)), t                              # same set of Gaussians per tile
        d = np.array([key_of[int(x)] for x in mine], np.int64)
        assert np.all(np.diff(d) >= 0), t                                                  # in depth order (ties in any order)
    if 'final_T_tiles' in dec:
        fT = tiles_to_image(dec['final_T_tiles'], width, height)
        assert float(f['final_T'].min()) > 1e-3                                            # the scene's premise: nobody terminated
        assert np.abs(fT.reshape(-1) - f['final_T']).max() < 1e-5


def wide_image_scene(n: int = 3000, seed: int = 22):      # seed 22: no two visible Gaussians share a depth key (tied keys keep K1's atomic compaction order on hardware)
    """3 000 Gaussians spread over a 20 000 x 36 px image (1 250 x 3 tiles): a fifth of them start at tile column >= 1024, beyond what a footprint row's
    10-bit box origin holds (csrc/fgs_math.h) -- they travel as escape rows and are re-tested by the instance kernel from the record."""
    from harness.scenes import View, make_s0
    p, v = make_s0(seed=seed, n=n)
    p['means'][:, 0] = p['means'][:, 0] * 60.0
    p['means'][:, 1] = p['means'][:, 1] * 0.02
    return p, View(v.w2c, v.position, 20000, 36, 660.0, 660.0, 10000.0, 18.0, 0.2, 1e4, torch.zeros(3))


def many_big_footprints_scene(n: int = 700, seed: int = 23):
    """700 large, faint Gaussians at 480 x 270 (690 tiles): ~450 of them have boxes of more than 256 candidate tiles, all inside ONE workgroup of the depth
    sort's last pass -- more than the 256 big footprints such a workgroup collects in LDS before it appends to the list directly (radix_sort.hip)."""
    from harness.scenes import View, make_s0
    p, v = make_s0(seed=seed, n=n)
    p['scales'] = p['scales'] + 2.6
    p['opacities'] = p['opacities'] - 3.0
    return p, View(v.w2c, v.position, 480, 270, 300.0, 300.0, 240.0, 135.0, 0.2, 1e4, torch.zeros(3))


DEPTH_RANGE_CASES = [(0.0, 1e4, 1.0, 4), (0.2, 1e4, 1.0, 3), (3.99, 4.01, 0.009, 2), (4.0, 4.00001, 0.0, 1)]      # near, far, z scale of the scene, sort passes


def depth_range_scene(near: float, far: float, zscale: float, n: int = 800, seed: int = 31):
    """S0-like scene squeezed in depth so that every Gaussian stays inside [near, far]: the depth sort orders key - bits(near) in ceil(bits / 9)
    passes -- 4 (near = 0: 31 bits), 3 (the default planes), 2, and ONE pass, which is the first (it makes up the values) and the last (it gathers the
    footprint rows) at once. zscale 0 puts every Gaussian at the same depth: 799 tied keys."""
    from harness.scenes import View, make_s0
    p, v = make_s0(seed=seed, n=n)
    p['means'][:, 2] = p['means'][:, 2] * zscale
    return p, View(v.w2c, v.position, v.width, v.height, v.focal_x, v.focal_y, v.center_x, v.center_y, near, far, torch.zeros(3))


def check_lists_up_to_ties(dec: dict, f: dict) -> None:
    """Counts, tile counts, sorted depth keys, ranges and tile keys exactly; every tile's list as a SET, in non-decreasing depth (Gaussians with equal
    depth keys keep K1's atomic compaction order on hardware, which is not the oracle's)."""
    assert dec['V'] == f['V'] and dec['I'] == f['I'] and np.array_equal(dec['n_touched'], f['n_touched'])
    sel = 1 if np.array_equal(dec['depth_keys1'], f['depth_keys']) else 0
    assert np.array_equal(dec[f'depth_keys{sel}'], f['depth_keys']) and np.array_equal(np.sort(dec[f'prim_idx{sel}']), np.sort(f['prim_idx']))
    assert np.array_equal(dec['ranges'], f['ranges']) and np.array_equal(dec['inst_keys'], f['inst_keys'])
    key_of = dict(zip(f['prim_idx'].tolist(), f['depth_keys'].tolist()))
    for t, (a, b) in enumerate(f['ranges']):
        mine = dec['inst_prims'][a:b]
        assert np.array_equal(np.sort(mine), np.sort(f['inst_prims'][a:b])), t
        assert np.all(np.diff(np.array([key_of[int(x)] for x in mine], np.int64)) >= 0), t


# ---- the two blend kernels against the oracle ON THE DEVICE'S OWN RECORDS (round 6) --------------------------------------------------------------------------
def check_blend_on_device_records(be, oracle, params, view, K=16, aa=False, device='cpu', label='', tol=1e-4, near_tol=1e-2, max_masked=3e-3):
    """K10 and K11 isolated from K1: the backend's forward pass runs as usual; its per-Gaussian records (mean2d, conic, opacity, colour -- decoded from the
    primitive blob) replace the oracle's in a second run of the oracle's blend (oracle.reblend) over the same instance lists, and the image, the final
    transmittances, the last contributors and K11's nine per-Gaussian sums (read back from the blob behind fgs_backward) are held to `tol` outside the
    threshold-risk masks of THAT run. What the end-to-end comparisons cannot separate -- a needle-shaped Gaussian whose conic differs by 2e-3 between two fp32
    evaluations of kf:140-150 moves its pixels on both sides here."""
    import torch
    S, RS = settings_pair(view, K, aa, device=device)
    dp = {k: v.to(device) for k, v in params.items()}
    n = dp['means'].shape[0]
    res = be.forward(*[dp[k] for k in NAMES], RS)
    f = oracle.forward(*np_params(params), S, bucket_size=64)
    dec = decode_forward(be, res, n, view.width, view.height)
    # the DEVICE's discrete structure as well (tile counts, bounds, instance lists, ranges, buckets): equal depth keys keep K1's arrival order there and a
    # cull / bound on its threshold may fall the other way -- neither is the blend kernels' business
    T_ = f['T']
    fdev = dict(f)
    if 'B' not in dec:                                       # nothing visible: no bucket was laid out
        dec = dict(dec, B=0, bucket_offsets=np.zeros(T_, np.uint32))
    fdev.update(I=dec['I'], B=dec['B'], n_touched=np.ascontiguousarray(dec['n_touched']), screen_bounds=np.ascontiguousarray(dec['screen_bounds']),
                inst_prims=np.ascontiguousarray(dec['inst_prims']), ranges=np.ascontiguousarray(dec['ranges'][:T_]),
                bucket_offsets=np.ascontiguousarray(dec['bucket_offsets'][:T_]))
    vis = dec['n_touched'] > 0
    f2 = oracle.reblend(fdev, S, np.ascontiguousarray(dec['mean2d'], np.float32), np.ascontiguousarray(dec['conic_opacity'], np.float32), np.ascontiguousarray(dec['color'], np.float32))
    masks = flip_masks(oracle, f2, S)
    pm, prim, near = masks['pixel'], masks['prim'], masks['near']
    report = {'masked_pixels': float(pm.mean()), 'masked_gaussians': float(prim.mean()), 'near_gaussians': float(near.mean())}
    assert report['masked_pixels'] < max_masked and report['masked_gaussians'] < max_masked, (label, 'masked fraction', report)
    img = res.image.cpu().numpy()
    err = np.abs(img.astype(np.float64) - f2['image']).max(axis=0)
    report['image'] = float(np.where(pm, 0.0, err).max() / max(1.0, float(np.abs(f2['image']).max())))
    assert report['image'] < tol, (label, 'image on the device records', report)
    fT = tiles_to_image(dec['final_T_tiles'], view.width, view.height, fill=1.0)
    report['final_T'] = float(np.where(pm, 0.0, np.abs(fT - f2['final_T'].reshape(fT.shape))).max())
    assert report['final_T'] < tol, (label, 'final transmittance', report)
    npr = tiles_to_image(dec['n_processed_tiles'], view.width, view.height)
    report['last_contributor_differs'] = float((npr != f2['n_processed'].reshape(npr.shape))[~pm].mean())
    assert report['last_contributor_differs'] < 3e-4, (label, 'last contributor', report)
    # backward: the nine sums of K11 per Gaussian, from the blob (they live behind the forward pass's records; hot Gaussians are folded by the backward pass)
    gi = np.random.default_rng(3).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
    g2 = oracle.backward(f2, S, gi, np.zeros((2, n), np.float32))
    be.backward(torch.zeros(2, n, device=device), torch.from_numpy(gi).to(device), res.image, dp['means'], dp['scales'], dp['rotations'], dp['opacities'],
                dp['sh_coefficients_rest'], res.buffers, RS, res.state)
    lp = be.blob_layout(0, n, view.width, view.height, res.state[1], res.state[2])
    acc = be.view(res.buffers[0].cpu(), lp, 'acc', torch.float32).reshape(n, 9).numpy()
    ref = np.concatenate([g2['_grad_mean2d'], g2['_grad_conic'].T, g2['_grad_opacity_acc'].reshape(n, 1), g2['_grad_color_acc'].reshape(n, 3)], axis=1)
    keep = vis & ~prim & ~near
    t64 = None
    for c, name in enumerate(('mean2d.x', 'mean2d.y', 'conic.a', 'conic.b', 'conic.c', 'opacity', 'colour.r', 'colour.g', 'colour.b')):
        scale = (float(np.abs(ref[vis, c]).max()) if vis.any() else 0.0) + 1e-30
        report[name] = float(np.abs(acc[keep, c] - ref[keep, c]).max() / scale) if keep.any() else 0.0
        report[name + '_near'] = float(np.abs(acc[vis & near & ~prim, c] - ref[vis & near & ~prim, c]).max() / scale) if (vis & near & ~prim).any() else 0.0
        _log_tolerance('blend_on_device_records', report[name], acc[keep, c], ref[keep, c], tensor=name, label=label.replace(' ', '_'))
        if report[name] >= tol:
            # the largest entry of this sum over the scene is itself an ill-conditioned sum (a handful of visible Gaussians, or a screen-filling one whose
            # dx^2 factors reach 1e5): three-way against the same sums in double -- the device may be at most twice as far from them as the fp32 oracle is
            if t64 is None:
                t64 = oracle.blend_sums_f64(f2, S, gi)['sums']
            e_dev = float(np.abs(acc[keep, c] - t64[keep, c]).max() / (float(np.abs(t64[vis, c]).max()) + 1e-30))
            e_o32 = float(np.abs(ref[keep, c] - t64[keep, c]).max() / (float(np.abs(t64[vis, c]).max()) + 1e-30))
            report[name + '_vs_f64'] = (e_dev, e_o32)
            log_note('blend_records_three_way', f'{e_dev:.3e}', label=label.replace(' ', '_'), tensor=name, oracle32=f'{e_o32:.3e}', device_vs_oracle32=f'{report[name]:.3e}')
            assert e_dev <= max(tol, 2.0 * e_o32), (label, 'K11 sum ' + name + ' (three-way against fp64)', report)
        assert report[name + '_near'] < near_tol, (label, 'K11 sum ' + name + ' (behind a borderline pair)', report)
    return report


def check_records_against_f64(be, oracle, params, view, K=16, aa=False, device='cpu', label='', thresholds=(1e-6, 1e-5, 1e-4, 1e-3, 1e-2)):
    """K1 on its own, conditioning-aware (round 6; the complement of check_blend_on_device_records). Every record component of every Gaussian both runs find
    visible -- mean2d, the three conic entries, opacity, colour -- is measured against its fp64 value (oracle.records_f64), on the device and in the fp32 oracle.
    An entry-by-entry bar is meaningless here (two draws from the same rounding-error distribution differ by a factor of four in 15 % of the ill-conditioned
    entries): the DISTRIBUTIONS are compared, as the element-wise three-way bar of the gradients does -- for every threshold t the device may have at most
    1.25 x as many entries farther than t (relative) from the fp64 value as the fp32 oracle has, plus four standard deviations of that count, and its worst entry
    may be at most 4 x the oracle's worst. A needle-shaped Gaussian's conic (cov / det, det a cancelling difference) is up to 0.2 of itself off the fp64 value in
    BOTH fp32 runs at S1; 2e-3 apart between them is inside that. Returns the counts."""
    S, RS = settings_pair(view, K, aa, device=device)
    dp = {k: v.to(device) for k, v in params.items()}
    n = dp['means'].shape[0]
    res = be.forward(*[dp[k] for k in NAMES], RS)
    f = oracle.forward(*np_params(params), S, bucket_size=64)
    dec = decode_forward(be, res, n, view.width, view.height)
    t = oracle.records_f64(f, S)
    both = (f['n_touched'] > 0) & (dec['n_touched'] > 0)
    report = {'visible': int(both.sum()), 'visible_to_one_side_only': int(((f['n_touched'] > 0) ^ (dec['n_touched'] > 0)).sum())}
    assert report['visible_to_one_side_only'] <= max(2, n // 1000), (label, report)
    for name in ('mean2d', 'conic_opacity', 'color'):
        a, b, x = np.asarray(dec[name], np.float64)[both], np.asarray(f[name], np.float64)[both], t[name][both]
        if not a.size:
            continue
        mag = np.maximum(np.abs(x), 1e-30)
        e_dev, e_o32 = np.abs(a - x) / mag, np.abs(b - x) / mag
        counts = {th: (int((e_dev > th).sum()), int((e_o32 > th).sum())) for th in thresholds}
        report[name] = {'worst_device': float(e_dev.max()), 'worst_oracle32': float(e_o32.max()), 'beyond_threshold (device, oracle32)': counts, 'entries': int(a.size),
                        'differ_between_the_fp32_runs': int((a != b).sum())}
        log_note('records_vs_f64', f'{float(e_dev.max()):.3e}', label=label.replace(' ', '_'), record=name, oracle32=f'{float(e_o32.max()):.3e}', counts=str(counts).replace(' ', ''))
        assert report[name]['worst_device'] <= 4.0 * report[name]['worst_oracle32'] + 2e-6, (label, name, 'worst entry', report)
        for th, (n_dev, n_o32) in counts.items():
            assert n_dev <= 1.25 * n_o32 + 4.0 * max(n_o32, 1) ** 0.5 + 2, (label, name, f'entries farther than {th:g} from the fp64 value', report)
    return report
