"""Random image sizes through the fused L1 + DSSIM loss on the GPU against the oracle: both call forms (autograd: forward call then backward call; one call with the
gradient, whose partial sums are reduced by workgroup 0 of the backward filter kernel), sizes that are not multiples of the 32 x 32 / 64 x 32 tiles, single rows /
columns, images smaller than the 11 x 11 window. usage: python tools/loss_sweep.py [N_CASES] [SEED]"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers, test_loss as T
from oracle import oracle as O
from harness.loss import l1_dssim_loss
from FasterGSCudaBackend._backend import default_backend
O.build(); be = default_backend()
cases, seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)
edges = [(1, 1), (1, 300), (300, 1), (5, 7), (11, 11), (32, 32), (33, 65), (31, 63), (64, 64), (12, 2000), (2000, 12)]
worst = [0.0, 0.0, 0.0]
for c in range(cases):
    h, w = edges[c] if c < len(edges) else (int(np.exp(rng.uniform(0, np.log(1200)))), int(np.exp(rng.uniform(0, np.log(2000)))))
    x, y = T._pair(h, w, seed=100 + c)
    ol, _, _, og = O.l1_dssim(x, y)
    tx = torch.from_numpy(x).cuda().requires_grad_(True); ty = torch.from_numpy(y).cuda()
    loss = l1_dssim_loss(tx, ty); (2.0 * loss).backward()                       # autograd form: reduce kernel between the two filter kernels
    loss1, grad1, _ = be.l1_dssim(tx.detach(), ty, 0.8, 0.2)                    # one call: the reduction rides in the backward kernel
    scale = np.abs(og).max() + 1e-30
    e = (abs(float(loss) - ol), float(np.abs(tx.grad.cpu().numpy() - 2.0 * og).max() / (2.0 * scale)), float(np.abs(grad1.cpu().numpy() - og).max() / scale))
    assert e[0] < 2e-6 and e[1] < 1e-4 and e[2] < 1e-4 and float(loss1) == float(loss), (h, w, e, float(loss1), float(loss))
    worst = [max(a, b) for a, b in zip(worst, e)]
    if (c + 1) % 25 == 0:
        print(f'{c + 1} sizes, last {h} x {w}: ok; worst so far: loss {worst[0]:.2e}, gradient (autograd form) {worst[1]:.2e}, gradient (one call) {worst[2]:.2e}', flush=True)
print(f'{cases} sizes: loss within 2e-6, gradients within 1e-4 of their maximum, the two call forms return the same loss bit for bit')
