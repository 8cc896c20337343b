"""TEST INFRASTRUCTURE: dry-runs a GPU tool script on the CPU simulation of the HIP library (tests/sim), to catch Python-level mistakes before
GPU minutes are spent on them. Installs the simulation backend as the process default, lets CPU tensors through the operator surface's device
guard and turns the torch.cuda bookkeeping calls the tools make into no-ops. Never imported by the product or by bench.py's measured paths.

usage: python tests/sim/run_with_sim.py tools/train_full.py --gt 3000 --points 400 --iters 60 --width 96 --height 72 ...
"""
import os, runpy, sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(REPO), str(REPO / 'faster-gaussian-splatting_amd'), str(REPO / 'tests')]
os.environ.setdefault('FGS_TOOL_DEVICE', 'cpu')
import torch
import helpers
import FasterGSCudaBackend
from FasterGSCudaBackend import _backend, aux_ops, rasterization

_backend._DEFAULT = helpers.sim_backend()
rasterization._require_gpu = lambda t: None
aux_ops._gpu = lambda t: None
from harness import densify as _densify
_densify._device_backend = lambda g, ops_backend=None: ops_backend or _backend._DEFAULT      # density control has no torch-op formulation: CPU tensors go to the simulation
for name in ('synchronize', 'reset_peak_memory_stats', 'empty_cache'):
    setattr(torch.cuda, name, lambda *a, **k: None)
for name in ('max_memory_allocated', 'max_memory_reserved'):
    setattr(torch.cuda, name, lambda *a, **k: 0)
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name='__main__')
