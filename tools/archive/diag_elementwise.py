#!/usr/bin/env python3
"""Where do the HIP gradients differ from the oracle ELEMENT by element (missing #3)? Runs on the GPU box: forward + backward of a
few small scenes through the HIP library (K11 variants 3 and 2) and the oracle, and saves both sets of gradients plus K11's raw accumulators to
gpurun_out/diag_elementwise_*.npz. The analysis against the fp64 autograd model (oracle/torch_check.py) happens off the box
(tools/diag_elementwise_analyze.py): which of the two fp32 results is closer to the truth, and in which term the difference sits."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / 'faster-gaussian-splatting_amd'), str(REPO / 'tests')]
import helpers  # noqa: E402
from harness.scenes import make_s0  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    O.build()
    from FasterGSCudaBackend._backend import default_backend
    be = default_backend()
    out = REPO / 'gpurun_out'
    out.mkdir(exist_ok=True)
    for tag, seed, n, shrink in (('a', 3, 300, 1.0), ('b', 11, 600, 0.3)):
        p, v = make_s0(seed=seed, n=n)
        p['means'][:, :2] *= shrink
        S, RS = helpers.settings_pair(v, 16, False, device='cuda')
        dp = {k: t.cuda().contiguous() for k, t in p.items()}
        f = O.forward(*helpers.np_params(p), S, bucket_size=64)
        gi = np.random.default_rng(3).standard_normal(f['image'].shape).astype(np.float32) / f['image'].size
        g = O.backward(f, S, gi)
        save = {f'in_{k}': p[k].numpy() for k in helpers.NAMES}
        save.update({f'oracle_{k}': g[k] for k in helpers.GRAD_KEYS})
        save['grad_image'] = gi
        save['oracle_image'] = f['image']
        for variant in (3, 2):
            be.lib.fgs_debug_set_backward_variant(variant)
            res = be.forward(*[dp[k] for k in helpers.NAMES], RS)
            grads = be.backward(torch.empty(0, device='cuda'), torch.from_numpy(gi).cuda(), res.image, dp['means'], dp['scales'], dp['rotations'],
                                dp['opacities'], dp['sh_coefficients_rest'], res.buffers, RS, res.state)
            torch.cuda.synchronize()
            for k, t in zip(helpers.GRAD_KEYS, grads):
                a = t.cpu().numpy().reshape(g[k].shape)
                save[f'hip{variant}_{k}'] = a
                print(tag, 'variant', variant, k, 'rel_inf %.2e' % helpers.rel_inf(a, g[k]), 'elem %.3e' % helpers.elementwise_fraction(a, g[k]))
            save[f'hip{variant}_image'] = res.image.cpu().numpy()
        be.lib.fgs_debug_set_backward_variant(3)
        np.savez_compressed(out / f'diag_elementwise_{tag}.npz', **save)


if __name__ == '__main__':
    main()
