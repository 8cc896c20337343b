"""GPU test (-m gpu) of `FasterGSCudaBackend._C`: the reference's own call pattern (torch_bindings/rasterization.py:43-110 -- an
autograd.Function that calls _C.forward, keeps the four opaque blobs + three integers on the ctx, and hands them back to _C.backward;
adam.py:27-36; rasterization.py:135-178 for inference / pruning_scores) replayed verbatim against the module, then compared with the
oracle. Nothing but the eight `_C` entry points of bindings.cpp:12-21 is used."""
from typing import Any

import numpy as np
import pytest
import torch
from torch.autograd.function import once_differentiable

import helpers
from harness.scenes import make_s0

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rasterize_class(_C):
    class _Rasterize(torch.autograd.Function):          # same statements as the reference's class, `_C` injected
        @staticmethod
        def forward(ctx: Any, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, densification_info, rasterizer_settings):
            (image, primitive_buffers, tile_buffers, instance_buffers, bucket_buffers,
             n_instances, n_buckets, instance_primitive_indices_selector) = _C.forward(
                means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, *rasterizer_settings.as_tuple())
            ctx.rasterizer_settings = rasterizer_settings
            ctx.buffer_state = (n_instances, n_buckets, instance_primitive_indices_selector)
            ctx.save_for_backward(image, means, scales, rotations, opacities, sh_coefficients_rest, primitive_buffers, tile_buffers,
                                  instance_buffers, bucket_buffers)
            ctx.densification_info = densification_info
            ctx.mark_non_differentiable(densification_info)
            return image

        @staticmethod
        @once_differentiable
        def backward(ctx: Any, grad_image):
            grads = _C.backward(ctx.densification_info, grad_image, *ctx.saved_tensors, *ctx.rasterizer_settings.as_tuple(), *ctx.buffer_state)
            return (*grads, None, None)
    return _Rasterize


def test_reference_call_pattern_through_c_module(hip_backend, oracle):
    from FasterGSCudaBackend import _C
    params, view = make_s0()
    S, RS = helpers.settings_pair(view, device=DEV)
    P = [torch.nn.Parameter(params[k].to(DEV)) for k in helpers.NAMES]
    dens = torch.zeros(2, 1000, device=DEV)
    image = _rasterize_class(_C).apply(*P, dens, RS)
    assert isinstance(image, torch.Tensor) and tuple(image.shape) == (3, view.height, view.width)
    gi = torch.randn(image.shape, generator=torch.Generator().manual_seed(0))
    (image * gi.to(DEV)).sum().backward()
    f = oracle.forward(*helpers.np_params(params), S)
    dens_o = np.zeros((2, 1000), np.float32)
    g = oracle.backward(f, S, gi.numpy(), dens_o)
    assert helpers.rel_inf(image.detach().cpu().numpy(), f['image']) < 1e-4
    for p, k in zip(P, helpers.GRAD_KEYS):
        assert helpers.rel_inf(p.grad.cpu().numpy().reshape(g[k].shape), g[k]) < 1e-4, k
    assert helpers.rel_inf(dens.cpu().numpy(), dens_o) < 1e-4
    # the three integers are exactly the reference's: instance count, bucket count, selector
    out = _C.forward(*[p.detach() for p in P], *RS.as_tuple())
    assert len(out) == 8 and all(t.dtype == torch.uint8 for t in out[1:5]) and all(isinstance(x, int) for x in out[5:])
    assert out[5] == f['I'] and out[7] in (0, 1)

    # FusedAdam.step of adam.py:11-36, one _C.adam_step per group
    lrs = [1.6e-4, 5e-3, 1e-3, 2.5e-2, 2.5e-3, 1.25e-4]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in P]
    ref = [(params[k].numpy().copy(), np.zeros(params[k].shape, np.float32), np.zeros(params[k].shape, np.float32)) for k in helpers.NAMES]
    with torch.no_grad():
        for step in (1, 2, 3):
            for p, (m, v), lr, (pp, mm, vv) in zip(P, state, lrs, ref):
                _C.adam_step(p.grad, p, m, v, step, lr, 0.9, 0.999, 1e-15)
                oracle.adam_step(np.ascontiguousarray(p.grad.cpu().numpy()), pp, mm, vv, step, lr)
    for p, (m, v), (pp, mm, vv) in zip(P, state, ref):
        assert helpers.rel_inf(p.detach().cpu().numpy(), pp) < 1e-6 and helpers.rel_inf(m.cpu().numpy(), mm) < 1e-6 and helpers.rel_inf(v.cpu().numpy(), vv) < 1e-6


def test_c_module_inference_and_pruning_scores(hip_backend, oracle):
    from FasterGSCudaBackend import _C
    params, view = make_s0()
    S, RS = helpers.settings_pair(view, bg=(0.3, 0.1, 0.9), device=DEV)
    dp = [params[k].to(DEV).contiguous() for k in helpers.NAMES]
    img = _C.inference(*dp, *RS.as_tuple(), False, True)                      # rasterization.py:135-156
    f = oracle.forward(*helpers.np_params(params), S, inference=True, to_chw=False, clamp_output=True)
    assert img.shape == f['image'].shape and helpers.rel_inf(img.cpu().numpy(), f['image']) < 1e-4
    scores = torch.zeros(1000, device=DEV)
    assert _C.pruning_scores(scores, *dp, *RS.as_tuple()) is None              # rasterization.py:159-178
    ref = np.zeros(1000, np.float32)
    oracle.pruning_scores(ref, *helpers.np_params(params), S)
    assert helpers.rel_inf(scores.cpu().numpy(), ref) < 1e-4


def test_c_module_aux_entry_points(hip_backend, oracle):
    from FasterGSCudaBackend import _C
    rng = np.random.default_rng(1)
    n = 500
    op = torch.from_numpy(rng.uniform(0.05, 0.95, (n, 1)).astype(np.float32)).to(DEV)
    sc = torch.from_numpy(rng.uniform(0.01, 0.2, (n, 3)).astype(np.float32)).to(DEV)
    ns = torch.from_numpy(rng.integers(1, 9, n)).to(DEV)
    new_op, new_sc = _C.relocation_adjustment(op, sc, ns)                       # densification.py:11
    r_op, r_sc = oracle.relocation_adjustment(op.cpu().numpy(), sc.cpu().numpy(), ns.cpu().numpy())
    assert helpers.rel_inf(new_op.cpu().numpy(), r_op) < 1e-5 and helpers.rel_inf(new_sc.cpu().numpy(), r_sc) < 1e-5
    means = torch.zeros(n, 3, device=DEV)
    noise = torch.randn(n, 3, device=DEV)
    rs, rq, ro = torch.randn(n, 3, device=DEV) * 0.3 - 3.0, torch.randn(n, 4, device=DEV), torch.randn(n, 1, device=DEV)
    assert _C.add_noise(rs, rq, ro, noise, means, 1e-3) is None                 # densification.py:21
    ref = np.zeros((n, 3), np.float32)
    oracle.add_noise(rs.cpu().numpy(), rq.cpu().numpy(), ro.cpu().numpy(), noise.cpu().numpy(), ref, 1e-3)
    assert helpers.rel_inf(means.cpu().numpy(), ref) < 1e-4
