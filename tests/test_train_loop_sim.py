"""CPU test of the whole training LOOP (random initialisation + carving, the schedule's callbacks, render / loss / backward / Adam through the
public operators, evaluation on held-out views) on the tests/sim build of the HIP sources: tools/train_full.py, the script that produces
profiles/r04_train_full*.json on the GPU, dry-run at toy size through tests/sim/run_with_sim.py. Both density policies of fastergs_garden.yaml:
ADC (the configuration's default, Model.py:312-366) and USE_MCMC (Model.py:367-457, Trainer.py:120-165,199). Checks control flow and
bookkeeping, not image quality."""
import json
import math
import subprocess
import sys

import pytest

import helpers

TOOL = [sys.executable, str(helpers.REPO / 'tests' / 'sim' / 'run_with_sim.py'), str(helpers.REPO / 'tools' / 'train_full.py')]
TOY = ['--gt', '1500', '--points', '300', '--iters', '16', '--width', '48', '--height', '36', '--schedule-scale', '0.002', '--eval-at', '16', '--ring-size', '2']


def _run(extra):
    helpers.sim_backend()
    r = subprocess.run(TOOL + TOY + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])


@pytest.mark.parametrize('policy', ['adc', 'mcmc'])
def test_training_loop_end_to_end_on_the_simulation(policy):
    d = _run(['--policy', policy] + (['--max-primitives', '420'] if policy == 'mcmc' else []))
    assert d['policy'] == policy and d['iterations_done'] == 16 and d['nonfinite_loss_windows'] == 0
    first, last = d['psnr']['0'], d['psnr']['16']
    assert all(math.isfinite(last[k]) for k in ('train_psnr_db', 'held_out_psnr_db'))
    assert d['active_sh_degree'] == 3                                           # the SH schedule ran (interval 2 at this scale)
    if policy == 'mcmc':
        assert last['train_psnr_db'] > first['train_psnr_db']                   # no opacity resets under MCMC: 16 iterations from grey blobs improve the images
        assert d['gaussians_end'] == 420 and d['gaussians_max'] == 420          # 5 % per densification step up to MAX_PRIMITIVES, never beyond
        counts = [c for _, c in d['count_curve_every_10th_call']]
        assert counts == sorted(counts)                                         # relocation replaces dead Gaussians, it never shrinks the set
    else:
        assert d['gaussians_end'] != d['gaussians_after_carving']               # density control changed the set
