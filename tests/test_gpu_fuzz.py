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
    budget = max(1e-2, 6.0 / min(n, pixels))
    # ... and the Gaussians that merely contribute to a pixel with such a pair are held to 1e-2 (helpers.check_flip_aware, near_tol)
    _flip_aware_forward_backward(hip_backend, oracle, p, view, label, adam_steps=2, K=K, aa=aa, max_masked=budget, near_tol=1e-2)


@pytest.mark.parametrize('seed', _SEEDS[::3])
def test_random_configuration_fused_backward_adam(hip_backend, oracle, seed):
    """The same configurations through fgs_backward_adam_fused (two steps): parameters and both moments against oracle backward -> oracle Adam."""
    from test_gpu_fused import _run
    p, view, K, aa, label = _configuration(seed)
    n = p['means'].shape[0]
    _run(hip_backend, oracle, p, view, K=K, aa=aa, steps=2, label=label, masked_budget=max(2e-2, 6.0 / n))
