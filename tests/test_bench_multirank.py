"""CPU test of bench.py's MULTI-RANK branch (the batch construction of dp_step, the second exchange, the wire-byte model, the roster all-gather,
the exposed-communication field, the max-over-ranks timing) before an 8-GPU node ever runs it: `bench.py --gpus N --sim` goes through the same
self-launch path (`python -m torch.distributed.run`, one process per rank) with the tests/sim build of the HIP sources as backend, CPU tensors and
gloo instead of RCCL. What is checked is the script's control flow and its output contract, not a number."""
import json
import os
import subprocess
import sys

import pytest

import helpers

BENCH = str(helpers.REPO / 'bench.py')
COMMON = ['--sim', '--scene', 'S0', '--n-gaussians', '300', '--steps', '1', '--warmup', '0', '--blocks', '2', '--watchdog', '600']


def _run(extra, env=None, timeout=900):
    helpers.sim_backend()        # build the simulation library once, here, instead of in N racing ranks
    return subprocess.run([sys.executable, BENCH] + extra + COMMON, capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, **(env or {})))


@pytest.mark.parametrize('n,mode', [(2, 'sharded'), (2, 'allreduce'), (8, 'sharded')])      # (zero1 shares allreduce's branch of this script: tests/test_distributed.py drives it)
def test_bench_multirank_branch_runs_and_reports_every_rank(n, mode):
    r = _run(['--gpus', str(n), '--dp-mode', mode])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:1000]                      # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    cfg = d['config']
    assert d['n_gpus'] == n and cfg['world'] == n and d['scaling'] == 'weak' and d['steps'] == 1
    assert 'SIMULATION' in d['data'] and cfg['backend'] == 'gloo'          # nobody can mistake this line for a measurement
    assert sorted(x['rank'] for x in cfg['ranks']) == list(range(n))       # every rank took part, each exactly once
    assert cfg['dp_mode'] == mode and f'dp{n}' in cfg['parallelism']
    assert cfg['wire_bytes_per_rank_per_step'] > 0
    assert (cfg['exposed_comm_ms_per_step'] is not None) == (mode == 'sharded')      # only the sharded step brackets its exchanges
    if mode == 'sharded':
        assert sum(x['n_gaussians_on_rank'] for x in cfg['ranks']) == 300   # the shards partition the scene
    else:
        assert all(x['n_gaussians_on_rank'] == 300 for x in cfg['ranks'])   # replicated parameters
    other = d['other_exchange']                                             # the second exchange ran as well
    assert other['dp_mode'] == ('allreduce' if mode == 'sharded' else 'sharded') and other['iters_per_sec'] > 0
    assert other['wire_bytes_per_rank_per_step'] > 0
    dry = d['dry_exchange']                                                 # one dry exchange of each kind at the step's sizes, next to the predicted wire time
    assert len(dry['devices']) == n and dry['allreduce_gradient_arena']['bytes'] == 236.0 * 300
    assert dry['allreduce_gradient_arena']['measured_ms'] > 0 and dry['allreduce_gradient_arena']['predicted_wire_ms'] > 0
    assert dry['sharded_all_to_all']['measured_ms_records'] > 0 and dry['sharded_all_to_all']['predicted_wire_ms'] > 0 and 'host memory' in dry['note']
    assert d['value'] > 0 and abs(d['value'] - n * 1e3 / d['ms_per_step']) < 1e-6 * d['value']      # whole-job rate = N views per step
    assert len(d['repeatability']['ms_per_step']) == 2
    assert d['roofline']['bound'] == 'hbm' and 'cpu_baseline' not in d     # N > 1: no CPU leg


def test_bench_multirank_a_failing_rank_fails_the_job_and_its_stderr_is_relayed():
    r = _run(['--gpus', '2'], env={'FGS_BENCH_SIM_FAIL_RANK': '1'}, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ''                                           # no line at all rather than a wrong one
    assert 'failure injected by FGS_BENCH_SIM_FAIL_RANK' in r.stderr and 'stderr.log' in r.stderr     # the rank's own traceback, relayed by the launcher


def test_bench_multirank_falls_back_to_allreduce_when_the_default_exchange_raises():
    """First contact with the fabric must end with a number: an exchange that RAISES on its first steps is replaced by north star's all-reduce
    on every rank, and the line says so."""
    r = _run(['--gpus', '2', '--dp-mode', 'sharded'], env={'FGS_BENCH_SIM_FAIL_MODE': 'sharded'}, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert d['dp_fallback']['from'] == 'sharded' and d['dp_fallback']['to'] == 'allreduce' and 'FGS_BENCH_SIM_FAIL_MODE' in d['dp_fallback']['error_on_this_rank']
    assert d['config']['dp_mode'] == 'allreduce' and 'allreduce' in d['config']['parallelism'] and d['n_gpus'] == 2 and d['value'] > 0
    assert all(x['n_gaussians_on_rank'] == 300 for x in d['config']['ranks'])      # replicated parameters: the all-reduce step ran
    assert 'other_exchange' not in d                                              # the failed mode is not tried a second time
