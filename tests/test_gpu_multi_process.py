"""GPU test (-m gpu) of the multi-rank trainers with MORE THAN ONE RANK on the real HIP kernels. The test box has one MI355X and RCCL refuses
two ranks on one device, so the two processes share cuda:0 and talk through `gloo` (which stages device tensors through host memory): the
exchange is not xGMI, but everything else is the shipped multi-GPU step -- two processes, torch.distributed with device tensors, uneven
all_to_all_single splits, the count all-gather, libfgs_hip.so's sharded entry points on records that crossed a process boundary. The result
must equal one process taking one Adam step on the sum of the two views' gradients (the same reference as the CPU gloo tests
tests/test_distributed.py / tests/test_sharded.py, here computed by the same HIP library). World 1 over RCCL itself: tests/test_gpu_rccl.py."""
import os
import socket
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from test_distributed import LRS, _setup_paths

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
STEPS = 2


def _device_scene(world=2):
    """tests/test_distributed.py's scene (200 Gaussians, 48 x 36) on the device, one view and one target per rank."""
    _setup_paths()
    from harness.scenes import View, make_s0
    params, v0 = make_s0(seed=5, n=200)
    settings = []
    for shift in [0.6 * i / (world - 1) for i in range(world)]:
        w2c = v0.w2c.clone()
        w2c[0, 3] = shift
        view = View(w2c, torch.tensor([-shift, 0.0, -4.0]), 48, 36, 48.0, 48.0, 24.0, 18.0, 0.2, 1e4, torch.zeros(3))
        settings.append(helpers.settings_pair(view, device=DEV)[1])
    dp = {k: v.to(DEV).contiguous() for k, v in params.items()}
    dt = [torch.full((3, 36, 48), 0.3 + 0.2 * i / (world - 1), device=DEV) for i in range(world)]
    return dp, settings, dt


def _worker(rank, world, mode, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _setup_paths()
        import FasterGSCudaBackend  # noqa: F401  (fails loudly if libfgs_hip.so is missing)
        from FasterGSCudaBackend._backend import default_backend
        from harness.distributed import SEGMENTS, ViewParallelTrainer
        from harness.sharded import ShardedTrainer, shard_of
        be = default_backend()
        dp, ds, dt = _device_scene(world)
        if mode.startswith('sharded'):
            tr = ShardedTrainer(be, shard_of(dp, rank, world), LRS, fused=(mode == 'sharded'))
            for _ in range(STEPS):
                tr.step(ds, dt[rank])
            full = tr.gather_parameters()
            out = {'full': {k: v.cpu() for k, v in full.items()}, 'shard': {k: v.cpu() for k, v in tr.params.items()},
                   'info': tr.densification_info.cpu(), 'counts': tr.last_counts.cpu()}
        else:
            tr = ViewParallelTrainer(be, dp, LRS, mode=mode)
            for _ in range(STEPS):
                tr.step(ds[rank], dt[rank])
            out = {'full': {k: tr.params[k].cpu() for k in SEGMENTS}, 'info': tr.gather_densification_info().cpu()}
        torch.cuda.synchronize()
        torch.save(out, Path(out_dir) / f'{mode}_{rank}.pt')
    finally:
        dist.destroy_process_group()


def _reference(be, world):
    """One process, replicated parameters: the views' gradients (each scaled 1 / world) summed, one Adam launch per step."""
    from harness.distributed import SEGMENTS, ViewParallelTrainer
    dp, ds, dt = _device_scene(world)
    tr = ViewParallelTrainer(be, dp, LRS)          # no process group: used for its arena / Adam plumbing only
    for _ in range(STEPS):
        tr.step_count += 1
        total = torch.zeros_like(tr.grad_arena)
        for s, t in zip(ds, dt):
            tr._render_backward(s, lambda img: (1.0 / world) * tr.image_gradient(img, t), True)
            total += tr.grad_arena
        tr.grad_arena.copy_(total)
        tr._adam(0, tr.param_arena.numel(), 0)
    torch.cuda.synchronize()
    return {k: tr.params[k].cpu() for k in SEGMENTS}, tr.densification_info.cpu(), {k: v.cpu() for k, v in dp.items()}


@pytest.mark.parametrize('mode,world', [('sharded', 2), ('sharded_unfused', 2), ('allreduce', 2), ('zero1', 2), ('sharded', 5), ('zero1', 3)])
def test_processes_sharing_the_device_on_the_hip_kernels_equal_the_summed_gradient_step(hip_backend, tmp_path, mode, world):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, mode, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f'{mode}_{i}.pt') for i in range(world)]
    ref, ref_info, start = _reference(hip_backend, world)
    for k in ref:
        assert all(torch.equal(r[0]['full'][k], r[i]['full'][k]) for i in range(1, world)), k      # every rank ends with the same model, bit for bit
        step_ref = (ref[k] - start[k]).numpy()
        assert abs(step_ref).max() > 0
        # float atomics arrive in another order: compare the step taken, relative to the largest step of the tensor
        assert helpers.rel_inf((r[0]['full'][k] - start[k]).numpy(), step_ref) < 1e-4, (mode, k)
        if mode.startswith('sharded'):
            for i in range(world):
                assert torch.equal(r[i]['shard'][k], r[0]['full'][k][i::world]), k   # rank i owns Gaussians i, i + world, ...
    if mode.startswith('sharded'):
        for i in range(world):      # owners accumulate the densification statistics of ALL views: no collective needed
            assert helpers.rel_inf(r[i]['info'].numpy(), ref_info[:, i::world].numpy()) < 1e-4
        assert all(torch.equal(r[0]['counts'], r[i]['counts']) for i in range(1, world)) and int(r[0]['counts'][..., 0].sum()) > 0
    else:
        assert helpers.rel_inf(r[0]['info'].numpy(), ref_info.numpy()) < 1e-4


@pytest.mark.parametrize('mode', ['sharded', 'allreduce'])
def test_bench_two_ranks_sharing_the_device_run_the_multi_rank_branch_on_the_real_kernels(mode):
    """bench.py --gpus 2 --shared-device: the script's N > 1 branch (self-launch through torch.distributed.run, dp_step batches, the second
    exchange, roster, wire bytes, exposed communication, max-over-ranks timing) with libfgs_hip.so and device tensors -- what
    tests/test_bench_multirank.py checks on the CPU simulation. S1 (1 M Gaussians at 1080p) so that launch sizes are real."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    cmd = [sys.executable, str(helpers.REPO / 'bench.py'), '--gpus', '2', '--shared-device', '--dp-mode', mode, '--scene', 'S1', '--steps', '3', '--warmup', '1',
           '--blocks', '2', '--watchdog', '400']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:1000]
    d = json.loads(lines[0])
    cfg = d['config']
    assert d['n_gpus'] == 2 and cfg['world'] == 2 and d['scaling'] == 'weak'
    assert 'SHARE cuda:0' in d['data'] and cfg['backend'] == 'gloo' and cfg['rccl_version'] is None       # nobody can mistake this line for a measurement
    assert sorted(x['rank'] for x in cfg['ranks']) == [0, 1] and all('MI3' in x['device'] or 'AMD' in x['device'] for x in cfg['ranks'])
    assert cfg['dp_mode'] == mode and cfg['wire_bytes_per_rank_per_step'] > 0
    assert (cfg['exposed_comm_ms_per_step'] is not None) == (mode == 'sharded')
    if mode == 'sharded':
        assert sum(x['n_gaussians_on_rank'] for x in cfg['ranks']) == cfg['n_gaussians']
    other = d['other_exchange']
    assert other['dp_mode'] == ('allreduce' if mode == 'sharded' else 'sharded') and other['iters_per_sec'] > 0
    assert d['value'] > 0 and abs(d['value'] - 2 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    assert d['roofline']['bound'] == 'hbm' and d['roofline']['achieved'] > 0 and 'cpu_baseline' not in d
