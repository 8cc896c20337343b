"""CPU test of the multi-GPU path: world_size-2 `gloo` processes run ViewParallelTrainer (allreduce and zero1 modes) on
the simulation backend and must end with identical parameters on both ranks, equal to a single-process run that sums the
two views' gradients itself (the only cross-rank semantic there is: SURVEY.md 8e)."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
LRS = {'means': 1.6e-4, 'sh_coefficients_0': 2.5e-3, 'sh_coefficients_rest': 1.25e-4, 'opacities': 2.5e-2, 'scales': 5e-3, 'rotations': 1e-3}


def _setup_paths():
    for p in (REPO, REPO / 'faster-gaussian-splatting_amd', REPO / 'tests'):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))


def _scene():
    _setup_paths()
    import helpers
    from harness.scenes import View, make_s0
    params, v0 = make_s0(seed=5, n=200)
    views = []
    for shift in (0.0, 0.6):
        w2c = v0.w2c.clone()
        w2c[0, 3] = shift
        views.append(View(w2c, torch.tensor([-shift, 0.0, -4.0]), 48, 36, 48.0, 48.0, 24.0, 18.0, 0.2, 1e4, torch.zeros(3)))
    settings = [helpers.settings_pair(v)[1] for v in views]
    targets = [torch.full((3, 36, 48), 0.3 + 0.2 * i) for i in range(2)]
    return params, settings, targets


def _worker(rank, world, mode, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _setup_paths()
        import helpers
        from harness.distributed import ViewParallelTrainer
        params, settings, targets = _scene()
        if mode.endswith('+old_backend'):
            # a backend like an older gloo: no reduce_scatter_tensor, and an all-gather that refuses an input aliasing its output. The trainer's probe
            # (ViewParallelTrainer._probe_collectives) must find both out at construction and fall back by itself
            genuine_gather = dist.all_gather_into_tensor

            def no_reduce_scatter(*a, **k):
                raise NotImplementedError('this backend has no reduce_scatter_tensor')

            def strict_gather(output, input, *a, **k):
                if output.untyped_storage().data_ptr() == input.untyped_storage().data_ptr():
                    raise RuntimeError('input aliases output')
                return genuine_gather(output, input, *a, **k)
            dist.reduce_scatter_tensor, dist.all_gather_into_tensor = no_reduce_scatter, strict_gather
        tr = ViewParallelTrainer(helpers.sim_backend(), params, LRS, mode=mode.split('+')[0], emulate_reduce_scatter=mode.endswith('+emulated'))
        if mode.endswith('+old_backend'):
            assert tr.emulate_reduce_scatter and tr.gather_from_copy
        elif mode == 'zero1':
            assert not tr.emulate_reduce_scatter and not tr.gather_from_copy          # the installed gloo does both: the calls RCCL gets are the ones tested
        for _ in range(2):
            tr.step(settings[rank], targets[rank])
        info = tr.gather_densification_info()
        torch.save({'params': {k: v.clone() for k, v in tr.params.items()}, 'info': info.clone()}, Path(out_dir) / f'{mode}_{rank}.pt')
    finally:
        dist.destroy_process_group()


def _single_process_reference():
    _setup_paths()
    import helpers
    from harness.distributed import SEGMENTS, ViewParallelTrainer
    params, settings, targets = _scene()
    be = helpers.sim_backend()
    tr = ViewParallelTrainer(be, params, LRS)          # world 1: used for its arena / Adam plumbing only
    info = torch.zeros(2, 200)
    for _ in range(2):
        tr.step_count += 1
        total = torch.zeros_like(tr.grad_arena)
        for s, t in zip(settings, targets):
            tr._render_backward(s, lambda img: 0.5 * tr.image_gradient(img, t), True)
            total += tr.grad_arena
        tr.grad_arena.copy_(total)
        tr._adam(0, tr.param_arena.numel(), 0)
    return {k: tr.params[k].clone() for k in SEGMENTS}, tr.densification_info.clone()


@pytest.mark.parametrize('mode', ['allreduce', 'zero1', 'zero1+emulated', 'zero1+old_backend'])
def test_view_parallel_world2_gloo(tmp_path, mode):
    """zero1 runs dist.reduce_scatter_tensor and the in-place dist.all_gather_into_tensor -- the calls RCCL gets on the GPUs (torch's gloo backend
    implements both); 'zero1+emulated' is the all-reduce form of the reduce-scatter, which must give the same parameters; 'zero1+old_backend' a backend without
    reduce_scatter_tensor and without aliased all-gather: the trainer's construction-time probe falls back by itself (round-5 advisor item)."""
    port = 29500 + (os.getpid() % 2000) + ('allreduce', 'zero1', 'zero1+emulated', 'zero1+old_backend').index(mode)
    mp.spawn(_worker, args=(2, mode, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / f'{mode}_0.pt')
    r1 = torch.load(tmp_path / f'{mode}_1.pt')
    ref_params, ref_info = _single_process_reference()
    for k in ref_params:
        assert torch.equal(r0['params'][k], r1['params'][k]), k                      # replicas stay bit-identical
        assert torch.allclose(r0['params'][k], ref_params[k], rtol=0, atol=1e-6), k   # == summed-gradient update
        assert (r0['params'][k] - _scene()[0][k]).abs().max() > 0
    assert torch.equal(r0['info'], r1['info']) and torch.allclose(r0['info'], ref_info, rtol=1e-5, atol=1e-7)
