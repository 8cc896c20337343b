"""update_3d_filter / relocation_adjustment / add_noise (reference filter3d.cu, kernels_mcmc.cuh):
simulation build vs the numpy restatement here, hardware under -m gpu."""
import numpy as np
import pytest
import torch

import helpers
from harness.scenes import make_s0


def _inputs(n=500, seed=0):
    p, v = make_s0(seed=seed, n=n)
    g = torch.Generator().manual_seed(seed)
    return p, v, g


def _check(be, oracle, dev):
    p, v, g = _inputs()
    n = p['means'].shape[0]
    # 3D filter
    filt = torch.full((n,), 1e3)
    filt[::7] = 1e-4                                   # already smaller than any new value: unchanged
    mask = torch.zeros(n, dtype=torch.bool)
    f_ref, m_ref = filt.numpy().copy(), mask.numpy().copy()
    args = (128, 96, 100.0, 110.0, 60.0, 50.0, 0.2, 0.15, 0.2 ** 0.5 / 100.0)
    oracle.update_3d_filter(p['means'].numpy(), v.w2c.numpy(), f_ref, m_ref, *args)
    f_d, m_d = filt.to(dev), mask.to(dev)
    be.update_3d_filter(p['means'].to(dev), v.w2c.to(dev), f_d, m_d, *args)
    assert np.array_equal(m_d.cpu().numpy(), m_ref) and helpers.rel_inf(f_d.cpu().numpy(), f_ref) < 1e-6
    assert 0 < m_ref.sum() < n
    # relocation
    op = torch.rand(n, 1, generator=g) * 0.9 + 0.05
    sc = torch.rand(n, 3, generator=g) * 0.1 + 0.01
    ns = torch.randint(0, 60, (n,), generator=g)
    o_ref, s_ref = oracle.relocation_adjustment(op.numpy(), sc.numpy(), ns.numpy())
    o_d, s_d = be.relocation_adjustment(op.to(dev), sc.to(dev), ns.to(dev))
    assert helpers.rel_inf(o_d.cpu().numpy(), o_ref) < 1e-5
    assert helpers.outlier_fraction(s_d.cpu().numpy(), s_ref, 1e-3, 1e-7) < 1e-2       # alternating binomial sum: ill-conditioned at n ~ 50
    # SGLD noise
    noise = torch.randn(n, 3, generator=g)
    rot = p['rotations'].clone()
    rot[3] = 0.0                                        # degenerate quaternion: mean untouched
    m_ref2 = p['means'].numpy().copy()
    oracle.add_noise(p['scales'].numpy(), rot.numpy(), p['opacities'].numpy(), noise.numpy(), m_ref2, 5e5 * 1.6e-4)
    m_d2 = p['means'].clone().to(dev)
    be.add_noise(p['scales'].to(dev), rot.to(dev), p['opacities'].to(dev), noise.to(dev), m_d2, 5e5 * 1.6e-4)
    assert helpers.rel_inf(m_d2.cpu().numpy(), m_ref2) < 1e-5 and np.array_equal(m_d2[3].cpu().numpy(), p['means'][3].numpy())
    assert np.abs(m_ref2 - p['means'].numpy()).max() > 0


def test_sim_aux_ops_match_numpy_restatement(sim_backend, oracle):
    _check(sim_backend, oracle, 'cpu')


@pytest.mark.gpu
def test_gpu_aux_ops_match_numpy_restatement(hip_backend, oracle):
    _check(hip_backend, oracle, 'cuda')
    import FasterGSCudaBackend as B                     # public names with the reference's signatures
    p, v, _ = _inputs(64)
    means = p['means'].cuda()
    before = means.clone()
    B.add_noise(p['scales'].cuda(), p['rotations'].cuda(), p['opacities'].cuda(), means, 80.0)
    assert not torch.equal(before, means)
    o, s = B.relocation_adjustment(torch.full((4, 1), 0.5).cuda(), torch.ones(4, 3).cuda(), torch.tensor([1, 2, 3, 4]).cuda())
    assert torch.allclose(o[0], torch.tensor(0.5, device='cuda')) and o.shape == (4, 1) and s.shape == (4, 3)
