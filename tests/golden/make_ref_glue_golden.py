"""Generates tests/golden/ref_glue_<scene>.npz and ref_glue_<scene>.trace.json by EXECUTING THE REFERENCE'S OWN PYTHON GLUE.

Build container only (needs /root/reference). The four files torch_bindings/{rasterization,adam,densification,filter3d}.py are loaded from
their path with importlib, unchanged -- never copied into this repository, never shipped -- and run on top of this repository's
`FasterGSCudaBackend._C` bound to the CPU simulation of the product sources (tests/sim, product flavour). The scenario (tests/ref_glue.py)
is this repository's own code and only uses the public operator names; what lands in the fixtures is data:

  *.npz        : images, gradients, densification statistics, parameters / Adam moments / step counts after each of three optimizer steps,
                 inference renders, pruning scores, relocation / noise / 3D-filter results -- as the reference's glue produced them;
  *.trace.json : every call that glue made into `_C` (bindings.cpp:12-21): entry point + per positional argument the scalar or the name of the
                 tensor handed over. This is the reference's argument routing as executed (rasterization.py:56-104 save_for_backward /
                 saved_tensors / as_tuple / buffer_state order, adam.py:27-36).

What this pins: rows a22 (_Rasterize glue), a29 (FusedAdam.step), a32 (RasterizerSettings) and the argument routing of every wrapper -- the
package's own operators must reproduce the fixture bit for bit on the same simulation (tests/test_ref_glue.py) and within 1e-4 on the MI355X
(tests/test_gpu_ref_glue.py). What it does NOT pin: the kernels' arithmetic -- both sides of the comparison run this repository's kernels
(the reference's are CUDA-only); that remains "parity unpinned" (oracle/fgs_oracle.c header).

Run from the repo root:  python tests/golden/make_ref_glue_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(REPO), str(REPO / 'faster-gaussian-splatting_amd'), str(REPO / 'tests')]
import ref_glue  # noqa: E402


def generate(name: str):
    """(arrays, trace) of scene `name` through the reference's glue on the simulation."""
    with ref_glue.simulated_backend():
        ops = ref_glue.reference_ops()
        from FasterGSCudaBackend import _C
        with ref_glue.Recorder(_C) as rec:
            arrays = ref_glue.run_scenario(ops, name, 'cpu', rec)
        return arrays, rec.trace()


if __name__ == '__main__':
    if not ref_glue.reference_available():
        sys.exit(f'{ref_glue.REFERENCE_BINDINGS} not found: this script runs in the build container only')
    for name in ref_glue.SCENES:
        arrays, trace = generate(name)
        np.savez_compressed(ref_glue.GOLDEN / f'ref_glue_{name}.npz', **arrays)
        (ref_glue.GOLDEN / f'ref_glue_{name}.trace.json').write_text(json.dumps(trace, indent=0) + '\n')
        size = (ref_glue.GOLDEN / f'ref_glue_{name}.npz').stat().st_size
        print(f'{name}: {len(arrays)} arrays ({size / 1e6:.2f} MB), {len(trace)} _C calls: ' + ' '.join(c['fn'] for c in trace))
