"""GPU test (-m gpu) of the RCCL code path: `torch.distributed` with backend "nccl" (= RCCL on ROCm), world size 1 on the one
GPU of the test box. Every collective the multi-GPU trainers issue (all_to_all_single with uneven splits, all_gather_into_tensor,
all_reduce, reduce_scatter_tensor) runs through RCCL here; the results must equal the single-GPU step of the same library
(reference semantics: one Adam step on the summed per-view gradients; Trainer.py:170-199 for the iteration).
World sizes > 1 are covered by the gloo tests (tests/test_distributed.py, tests/test_sharded.py) and measured by bench.py --gpus N."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import helpers
from harness.scenes import make_s0

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LRS = {'means': 1.6e-4, 'sh_coefficients_0': 2.5e-3, 'sh_coefficients_rest': 1.25e-4, 'opacities': 2.5e-2, 'scales': 5e-3, 'rotations': 1e-3}


@pytest.fixture(scope='module')
def rccl_world1():
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    assert dist.get_backend() == 'nccl'
    yield
    dist.destroy_process_group()


def _reference_steps(be, params, RS, target, steps):
    """The single-GPU step of the same library: forward, loss gradient, backward, one Adam launch over all groups."""
    from harness.distributed import SEGMENTS, ViewParallelTrainer
    saved = dist.is_initialized
    dist.is_initialized = lambda: False           # world-1 trainer without any collective
    try:
        tr = ViewParallelTrainer(be, params, LRS)
        helpers.seed_trainer_moments(tr, SEGMENTS)           # both sides start from the same non-zero moments (helpers.seeded_moments)
        for _ in range(steps):
            tr.step(RS, target)
    finally:
        dist.is_initialized = saved
    return {k: tr.params[k].clone() for k in SEGMENTS}, tr.densification_info.clone()


@pytest.mark.parametrize('mode', ['sharded', 'sharded_unfused', 'allreduce', 'zero1'])
def test_trainers_over_rccl_world1_equal_single_gpu_step(hip_backend, rccl_world1, mode):
    from harness.distributed import SEGMENTS, ViewParallelTrainer
    from harness.sharded import ShardedTrainer
    params, view = make_s0(n=3000)
    _, RS = helpers.settings_pair(view, device=DEV)
    dp = {k: v.to(DEV).contiguous() for k, v in params.items()}
    target = torch.full((3, view.height, view.width), 0.4, device=DEV)
    ref, ref_info = _reference_steps(hip_backend, dp, RS, target, 2)
    if mode.startswith('sharded'):
        tr = ShardedTrainer(hip_backend, dp, LRS, fused=(mode == 'sharded'))
        helpers.seed_trainer_moments(tr, SEGMENTS)
        for _ in range(2):
            tr.step([RS], target)
        got = tr.gather_parameters()
    else:
        tr = ViewParallelTrainer(hip_backend, dp, LRS, mode=mode)
        helpers.seed_trainer_moments(tr, SEGMENTS)
        for _ in range(2):
            tr.step(RS, target)
        got = {k: tr.params[k] for k in SEGMENTS}
        total = tr.gather_densification_info()
        assert torch.equal(total, tr.densification_info)          # world 1: the sum over ranks is the local tensor, which is left untouched
    torch.cuda.synchronize()
    for k in SEGMENTS:
        moved = (ref[k] - dp[k]).abs().max().item()
        assert moved > 0
        # float atomics in a different order: compare the step taken, relative to the largest step of the tensor
        assert helpers.rel_inf((got[k] - dp[k]).cpu().numpy(), (ref[k] - dp[k]).cpu().numpy()) < 1e-4, (mode, k)
    assert helpers.rel_inf(tr.densification_info.cpu().numpy(), ref_info.cpu().numpy()) < 1e-4


def test_bench_under_torchrun_world1_prints_one_json_line_of_the_single_gpu_iteration():
    """The driver launches bench.py for N > 1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`. With N = 1 the
    same launch must time the single-GPU iteration (the N = 1 point of a scaling curve IS the BENCH number), print exactly ONE line on
    stdout -- RCCL's version banner and everything else goes to stderr -- and name the ranks that took part."""
    import json
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(port),
           str(helpers.REPO / 'bench.py'), '--gpus', '1', '--scene', 'S0', '--steps', '3', '--warmup', '1', '--blocks', '2', '--no-extras',
           '--no-cpu-baseline', '--no-pmc']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['config']['parallelism'] == 'single GPU' and d['config']['world'] == 1
    assert d['config']['backend'] == 'nccl' and d['config']['rccl_version'] and len(d['config']['ranks']) == 1
    assert d['config']['ranks'][0]['rank'] == 0 and d['config']['ranks'][0]['n_gaussians_on_rank'] == 1000
    assert len(d['repeatability']['ms_per_step']) == 2 and d['peak_vram_GB']['peak_allocated_GB'] > 0
    assert d['roofline']['frac'] > 0 and 'stage_ms_per_step' in d
