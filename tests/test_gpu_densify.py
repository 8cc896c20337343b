"""GPU tests (-m gpu) of the maintenance passes of csrc/densify.hip against the numpy restatement of the
reference's Model.py:312-366 (adaptive density control with the optimizer-state surgery), :275-306 (prune / sort) and :459-463 (Morton
order) in oracle/oracle.py. Copies must be bit-exact; the split children's means / scales go through exp / log / sqrt of different
libms (2e-6)."""
import numpy as np
import pytest
import torch

import helpers  # noqa: F401
from test_densify import ORDER, _adc_case, check_adc_against_restatement

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('n,prune_large,with_state', [(700, True, True), (100_003, True, True), (100_003, False, False)])
def test_adaptive_density_control_on_device(hip_backend, oracle, n, prune_large, with_state):
    counts = check_adc_against_restatement(hip_backend, oracle, DEV, n=n, prune_large=prune_large, with_state=with_state)
    assert sum(counts) > 0


def test_gather_rows_and_morton_order_on_device(hip_backend, oracle):
    P, M, V, _ = _adc_case(n=200_001, device=DEV)
    order = hip_backend.morton_order(P['means'])
    assert np.array_equal(order.cpu().numpy(), oracle.morton_order(P['means'].cpu().numpy()))
    tensors = [P[k] for k in ORDER] + [M[k] for k in ORDER] + [V[k] for k in ORDER]
    outs = hip_backend.gather_rows(tensors, order)
    for t, o in zip(tensors, outs):
        assert torch.equal(o, t[order])
    keep = torch.nonzero(torch.rand(200_001, device=DEV) > 0.3).flatten()
    for t, o in zip(tensors, hip_backend.gather_rows(tensors, keep)):
        assert torch.equal(o, t[keep])


def test_training_with_device_densification(hip_backend):
    """harness.densify on device tensors takes the kernel path; a short run with clone / split / prune / Morton sort keeps training."""
    from harness import densify as D
    from harness import trainer as T
    from harness.scenes import make_s0
    params, view = make_s0(n=4000)
    g = T.Gaussians(params, DEV)
    g.training_setup(training_cameras_extent=5.0)
    v = view.to(DEV)
    target = torch.rand(3, view.height, view.width, generator=torch.Generator().manual_seed(3)).to(DEV)
    n0, losses = g.means.shape[0], []
    for it in range(60):
        if it in (20, 40):
            stats = D.adaptive_density_control(g, 1e-7, 0.005, False)
            assert stats['cloned'] + stats['split'] > 0 and g.means.is_cuda
            D.reset_densification_info(g)
            D.apply_morton_ordering(g)
            st = g.optimizer.state[g.means]
            assert st['exp_avg'].shape == g.means.shape and st['exp_avg'].is_cuda
        losses.append(float(T.training_iteration(g, v, target, it)))
    assert g.means.shape[0] != n0 and np.isfinite(losses).all() and losses[-1] < losses[0]
