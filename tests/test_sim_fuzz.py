"""CPU test: the seeded random configurations of tests/test_gpu_fuzz.py (ragged counts, odd image sizes, near / far planes, every SH degree, both
antialiasing modes, screen-filling / sub-pixel / culled / degenerate Gaussians, opacities on the 1/255 cut) through the product sources in the
fiber simulator against the oracle: every forward intermediate and the image BIT-EXACT (the simulator shares libm with the oracle and keeps
the reference's operation order), the gradients to 1e-5 (summation order)."""
import os

import pytest

import helpers
from test_sim_parity import _run, fused_equals_backward_then_adam


_SEEDS = range(*(int(x) for x in os.environ['FGS_FUZZ_SEEDS'].split('-'))) if os.environ.get('FGS_FUZZ_SEEDS') else range(32)


@pytest.mark.parametrize('seed', _SEEDS)
def test_random_configuration_in_the_simulator(sim_backend, oracle, seed):
    p, view, K, aa, label = helpers.fuzz_configuration(seed)
    _run(helpers.poisoned(sim_backend), oracle, p, view, K, aa)        # scratch buffers arrive as 0xFF bytes: nothing may depend on their contents


@pytest.mark.parametrize('seed', range(0, 32, 5))
def test_random_configuration_fused_equals_unfused_in_the_simulator(sim_backend, seed):
    """The same configurations: fgs_backward_adam_fused == fgs_backward -> fgs_adam_step_multi, two steps, bit for bit."""
    p, view, K, aa, label = helpers.fuzz_configuration(seed)
    fused_equals_backward_then_adam(helpers.poisoned(sim_backend), p, view, K, aa)
