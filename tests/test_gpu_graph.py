"""GPU test (-m gpu): one whole training iteration -- fgs_forward_async, the fused loss, fgs_backward, the one-launch Adam -- is free of host
synchronisation and can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed (the reference blocks the host
three times per forward pass, forward.cu:100,102,234). Replays must reproduce the eager iteration."""
import numpy as np
import pytest
import torch

import helpers
from harness.scenes import make_s0

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ORDER = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')


def _iteration(be, P, M, V, RS, target, capacity, step):
    res = be.forward(*[P[k] for k in helpers.NAMES], RS, instance_capacity=capacity)
    grad_image = be.l1_dssim(res.image, target, 0.8, 0.2)[1]
    grads = be.backward(None, grad_image, res.image, P['means'], P['scales'], P['rotations'], P['opacities'], P['sh_coefficients_rest'],
                        res.buffers, RS, res.state)
    gmap = dict(zip(helpers.NAMES, grads))
    be.adam_step_multi([gmap[k] for k in ORDER], [P[k] for k in ORDER], [M[k] for k in ORDER], [V[k] for k in ORDER], [step] * 6,
                       [1.6e-4, 2.5e-3, 1.25e-4, 2.5e-2, 5e-3, 1e-3], 0.9, 0.999, 1e-15)
    return res


def test_training_iteration_captures_into_a_graph(hip_backend):
    params, view = make_s0(n=5000)
    _, RS = helpers.settings_pair(view, device=DEV)
    target = torch.rand(3, view.height, view.width, generator=torch.Generator().manual_seed(1)).to(DEV)
    seeds = {k: helpers.seeded_moments(params[k].shape, 11 + i) for i, k in enumerate(ORDER)}   # non-zero moments: see helpers.seeded_moments
    fresh = lambda: ({k: params[k].to(DEV).clone() for k in ORDER}, {k: seeds[k][0].to(DEV) for k in ORDER},
                     {k: seeds[k][1].to(DEV) for k in ORDER})
    sync = hip_backend.forward(*[params[k].to(DEV) for k in helpers.NAMES], RS)
    capacity = int(1.25 * sync.state[1]) + 4096

    # eager reference: two iterations (the Adam step count is baked into a captured launch, so both sides use step = 1 twice)
    P, M, V = fresh()
    for _ in range(2):
        _iteration(hip_backend, P, M, V, RS, target, capacity, 1)
    torch.cuda.synchronize()
    ref = {k: P[k].clone() for k in ORDER}

    P, M, V = fresh()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                               # warm-up on the capture stream (allocator, lazy library state)
        Pw, Mw, Vw = fresh()
        _iteration(hip_backend, Pw, Mw, Vw, RS, target, capacity, 1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = _iteration(hip_backend, P, M, V, RS, target, capacity, 1)
    start = {k: params[k].to(DEV) for k in ORDER}
    for k in ORDER:                                             # capture does not execute: parameters are still the initial ones
        assert torch.equal(P[k], start[k]), k
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    host, event = hip_backend.forward_counts(res, 5000)
    event.synchronize()
    assert int(host[2]) == 0 and int(host[1]) > 0
    # Two eager runs of these two iterations already differ: the gradients carry last-bit noise from the order of K11's float atomics, and a
    # parameter whose update lands on a rounding tie comes out one ulp apart after the first step, up to three after the second
    # (tools/archive/eager_repeat.py: means of ONE Gaussian, 1.8e-7 = 3 ulp at 0.5, is the whole difference in 24 runs). So: the step each tensor took
    # agrees to 1e-4 of its largest step, plus four ulp of the parameter itself.
    for k in ORDER:
        moved = (ref[k] - start[k]).abs().max().item()
        diff = (P[k] - ref[k]).abs()
        bound = 1e-4 * moved + 4.0 * torch.finfo(torch.float32).eps * ref[k].abs()
        assert moved > 0 and bool((diff <= bound).all()), (k, float((diff - bound).max()), moved)
