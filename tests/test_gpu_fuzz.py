"""Seeded random configurations against the oracle on hardware: ragged Gaussian counts (1, 63, 64, 65 ... : last waves, last blocks of 64,
sort workgroups of one item), image sizes that are not multiples of the 16 x 12 tile or the 8 x 4 sub-tile, near / far planes that change
the depth-key range the sort works on, every SH degree, both antialiasing modes, screen-filling and sub-pixel Gaussians, Gaussians behind the
near plane and beyond the far plane, degenerate quaternions, opacities on the 1/255 cut. Same bar as the fixed-size tests: 1e-4 outside the
oracle's threshold-risk masks (the mask budget is at least a handful of entries: one risky pixel of a 17 x 13 image is more than 1e-3 of it)."""
import os

import numpy as np
import pytest
import torch

import helpers
from harness.scenes import View, make_s0
from test_gpu_parity import _flip_aware_forward_backward

pytestmark = pytest.mark.gpu


_configuration = helpers.fuzz_configuration
# FGS_FUZZ_SEEDS=a-b widens the sweep for an occasional deep run (default: the 32 seeds the suite has always used)
_SEEDS = range(*(int(x) for x in os.environ['FGS_FUZZ_SEEDS'].split('-'))) if os.environ.get('FGS_FUZZ_SEEDS') else range(32)


@pytest.mark.parametrize('seed', _SEEDS)
def test_random_configuration_against_oracle(hip_backend, oracle, seed):
    p, view, K, aa, label = _configuration(seed)
    n, pixels = p['means'].shape[0], view.width * view.height
    # the scenes are adversarial by construction -- one Gaussian in twenty sits on the opacity cut, so its whole footprint is at the alpha
    # threshold: the oracle's risk masks are allowed 1 % here (0.1 % in the fixed-size tests); outside them the bar is the same 1e-4
    budget = max(2e-2 if n <= 1000 else 1e-2, 6.0 / min(n, pixels))     # (2 % up to 1000 Gaussians: 8 of 500 in seed 4092 of a 4000-seed sweep, round 6; realised in the default seeds: <= 0.4 %)
    # ... and the Gaussians that merely contribute to a pixel with such a pair are held to 1e-2 (helpers.check_flip_aware, near_tol)
    _flip_aware_forward_backward(hip_backend, oracle, p, view, label, adam_steps=2, K=K, aa=aa, max_masked=budget, near_tol=1e-2)


@pytest.mark.parametrize('seed', _SEEDS[::3])
def test_random_configuration_fused_backward_adam(hip_backend, oracle, seed):
    """The same configurations through fgs_backward_adam_fused (two steps): parameters and both moments against oracle backward -> oracle Adam."""
    from test_gpu_fused import _run
    p, view, K, aa, label = _configuration(seed)
    n = p['means'].shape[0]
    _run(hip_backend, oracle, p, view, K=K, aa=aa, steps=2, label=label, masked_budget=max(3e-2 if n <= 1000 else 2e-2, 6.0 / n), near_tol=1e-2, skip_on_new_ties=True)


def _mid_scale_configuration(seed: int):
    """Garden-like scenes of 10^5 Gaussians seen from a random point of the orbit (radius, height, image size, focal length, SH degree,
    antialiasing mode, opacity level all drawn): the regime between the 10^3 scenes above and the fixed 10^6 ones of test_gpu_parity.py."""
    import math
    from harness.scenes import look_at_view, make_garden_like
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([100_003, 200_000, 300_001]))
    p = make_garden_like(n)
    p['opacities'] = p['opacities'] + float(rng.choice([0.0, -1.5, -3.0]))            # opaque surfaces ... deep semi-transparent layering
    W, H = int(rng.integers(320, 1280)), int(rng.integers(200, 720))
    a, radius, height = float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(2.5, 7.0)), float(rng.uniform(0.3, 3.0))
    view = look_at_view((radius * math.cos(a), -height, radius * math.sin(a)), (0.0, 0.0, 0.0), W, H, float(W) * float(rng.uniform(0.5, 1.2)))
    K, aa = int(rng.choice([1, 4, 9, 16])), bool(rng.integers(0, 2))
    return p, view, K, aa, f'mid-scale seed {seed}: n={n} {W}x{H} radius={radius:.2f} K={K} aa={aa}'


_MID_SEEDS = range(*(int(x) for x in os.environ['FGS_MID_SEEDS'].split('-'))) if os.environ.get('FGS_MID_SEEDS') else range(4)


@pytest.mark.parametrize('seed', _MID_SEEDS)
def test_random_mid_scale_configuration_against_oracle(hip_backend, oracle, seed):
    p, view, K, aa, label = _mid_scale_configuration(seed)
    # the index of a pixel's last contributor may differ outside the mask on 3e-4 of the pixels: with opacities lowered by 3 most lists end in a
    # run of pairs near the 1/255 cut, and a flip of the very last one moves the image by less than the bar (seed 4 of a 24-seed sweep: 1.3e-4)
    _flip_aware_forward_backward(hip_backend, oracle, p, view, label, adam_steps=2, K=K, aa=aa, max_masked=3e-3, last_contributor_budget=3e-4, near_tol=1e-2)


# Mid-scale seeds of a 600-seed sweep (round 6, profiles/r06_fuzz_sweeps.txt) whose END-TO-END comparison fails on a few dozen pixels because one needle-shaped
# Gaussian's conic (cov / det, det a cancelling difference) differs by up to 2e-3 between the device's K1 and the oracle's: with the device's own records on both
# sides the two blend kernels are held to the usual bars on exactly those scenes, next to the suite's own four.
@pytest.mark.parametrize('seed', sorted(set(_MID_SEEDS) | {90, 201, 264, 351, 469}))
def test_mid_scale_blend_kernels_on_the_device_records(hip_backend, oracle, seed):
    p, view, K, aa, label = _mid_scale_configuration(seed)
    helpers.check_blend_on_device_records(hip_backend, oracle, p, view, K, aa, device='cuda', label=label)


@pytest.mark.parametrize('seed', _SEEDS[::2])
def test_random_configuration_blend_kernels_on_the_device_records(hip_backend, oracle, seed):
    p, view, K, aa, label = _configuration(seed)
    n, pixels = p['means'].shape[0], view.width * view.height
    helpers.check_blend_on_device_records(hip_backend, oracle, p, view, K, aa, device='cuda', label=label, max_masked=max(2e-2, 6.0 / min(n, pixels)))


@pytest.mark.parametrize('seed', sorted(set(_MID_SEEDS) | {90, 201, 264, 351, 469}))
def test_mid_scale_records_against_fp64_conditioning_aware(hip_backend, oracle, seed):
    """... and K1 of the same scenes on its own: the records that break the end-to-end comparison of seeds 90 / 201 / 264 / 351 / 469 are as far from the fp64
    values in the fp32 oracle as they are on the device (helpers.check_records_against_f64)."""
    p, view, K, aa, label = _mid_scale_configuration(seed)
    helpers.check_records_against_f64(hip_backend, oracle, p, view, K, aa, device='cuda', label=label)
