"""pytest configuration: import paths, the `gpu` marker, shared fixtures.

CPU suite   : python -m pytest tests -x -q -m "not gpu"   (oracle, golden vectors, simulation parity, ABI, gloo)
GPU suite   : python -m pytest tests -x -q -m gpu          (HIP library through the C ABI vs the oracle)
"""
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / 'faster-gaussian-splatting_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    lib = REPO / 'faster-gaussian-splatting_amd' / 'libfgs_hip.so'
    if not lib.exists():   # hipcc cross-compiles gfx950 without a GPU; normally __graft_entry__.build() has done this already
        import subprocess
        subprocess.run(['make', '-C', str(lib.parent / 'csrc'), '-j8'], check=False, capture_output=True)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope='session')
def sim_backend():
    """Backend bound to the CPU simulation of the HIP library (tests/sim) -- test infrastructure, never the product."""
    import helpers
    return helpers.sim_backend()


@pytest.fixture(scope='session')
def sim_product_backend():
    """The simulation of the PRODUCT flavour: the same sources without -DFGS_DEV_SWITCHES (every A/B switch a constant, one formulation per kernel)."""
    import helpers
    return helpers.sim_backend(product=True)


@pytest.fixture(scope='session')
def hip_backend():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    import FasterGSCudaBackend  # noqa: F401  (fails loudly if libfgs_hip.so is missing)
    from FasterGSCudaBackend._backend import default_backend
    return default_backend()


@pytest.fixture(scope='session')
def hip_dev_backend():
    """Backend bound to libfgs_hip_dev.so: the product sources built with -DFGS_DEV_SWITCHES (K11's A/B variants, fgs_debug_set_option). Only the
    tests that compare formulations use it; everything else runs on the product library."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    import helpers
    _lib, _backend = helpers.backend_modules()
    if not _lib.DEV_LIBRARY.exists():
        import subprocess
        subprocess.run(['make', '-C', str(_lib.DEV_LIBRARY.parent / 'csrc'), '-j8', 'dev'], check=True, capture_output=True)
    return _backend.Backend(_lib.bind(_lib.DEV_LIBRARY))
