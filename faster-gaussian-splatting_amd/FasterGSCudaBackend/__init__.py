"""FasterGSCudaBackend -- drop-in package name of the reference's backend, implemented for AMD Instinct MI355X (gfx950).

Exports the names Renderer.py:16 and Model.py:17 import (reference FasterGSCudaBackend/__init__.py:13-18). The hot path
(diff_rasterize, rasterize, RasterizerSettings, FusedAdam) is implemented; the entry points the garden configuration
never calls (pruning scores, 3D filter, MCMC) raise NotImplementedError with that pointer.
"""
from ._lib import ExtensionError, library

library()   # fail loudly at import time if libfgs_hip.so is missing (reference: __init__.py:19-20)

from ._backend import RasterizerSettings  # noqa: E402
from .rasterization import diff_rasterize, rasterize  # noqa: E402
from .adam import FusedAdam  # noqa: E402
from .fused import FusedRasterizerOptimizer  # noqa: E402


def _out_of_scope(name: str):
    def fn(*_args, **_kwargs):
        raise NotImplementedError(f'{name} is outside the hot-path scope of this build (SURVEY.md 8f); '
                                  f'it is not used by fastergs_garden.yaml')
    fn.__name__ = name
    return fn


update_pruning_scores = _out_of_scope('update_pruning_scores')
update_3d_filter = _out_of_scope('update_3d_filter')
relocation_adjustment = _out_of_scope('relocation_adjustment')
add_noise = _out_of_scope('add_noise')

__all__ = ['diff_rasterize', 'rasterize', 'update_pruning_scores', 'RasterizerSettings', 'FusedAdam', 'update_3d_filter',
           'relocation_adjustment', 'add_noise', 'FusedRasterizerOptimizer', 'ExtensionError']
