"""FasterGSCudaBackend -- drop-in package name of the reference's backend, implemented for AMD Instinct MI355X (gfx950).

Exports the names Renderer.py:16 and Model.py:17 import (reference FasterGSCudaBackend/__init__.py:13-18). The hot path
(diff_rasterize, rasterize, RasterizerSettings, FusedAdam) plus the operators the garden configuration never calls
(update_pruning_scores, update_3d_filter, relocation_adjustment, add_noise: SURVEY.md section 8f) are all implemented.
"""
from ._lib import ExtensionError, library

library()   # fail loudly at import time if libfgs_hip.so is missing (reference: __init__.py:19-20)

from ._backend import RasterizerSettings  # noqa: E402
from .rasterization import (async_forward_scope, async_forward_stats, diff_rasterize, live_block_stats, rasterize, set_async_forward,  # noqa: E402
                            set_live_block_handover, take_async_overflow, update_pruning_scores)
from .adam import FusedAdam  # noqa: E402
from .fused import FusedRasterizerOptimizer  # noqa: E402


from .aux_ops import add_noise, relocation_adjustment, update_3d_filter  # noqa: E402

__all__ = ['diff_rasterize', 'rasterize', 'update_pruning_scores', 'RasterizerSettings', 'FusedAdam', 'update_3d_filter',
           'relocation_adjustment', 'add_noise', 'FusedRasterizerOptimizer', 'ExtensionError', 'set_async_forward', 'async_forward_stats',
           'set_live_block_handover', 'live_block_stats', 'take_async_overflow', 'async_forward_scope']
