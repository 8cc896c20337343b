"""CPU test of bench.py's output contract: ONE JSON line on stdout and nothing else, whatever the libraries underneath print (RCCL writes a
version banner to stdout when it creates a communicator: file descriptor 1 points at stderr during the run, bench.py:main). Exercised here
through the only leg that needs no GPU, --cpu-baseline-only (the oracle timed on the S0 scene)."""
import json
import subprocess
import sys

import helpers


def test_cpu_baseline_only_prints_exactly_one_json_line_whatever_the_libraries_print():
    """np.sign (called inside the timed oracle loop) is wrapped to print a banner, as a library would: it must not reach stdout."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--cpu-baseline-only', '--scene', 'S0']; "
            "import numpy as _np; _orig = _np.sign\n"
            "def noisy(*a, **k):\n    print('BANNER: not json'); return _orig(*a, **k)\n"
            "_np.sign = noisy\n"
            f"runpy.run_path({str(helpers.REPO / 'bench.py')!r}, run_name='__main__')")
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'BANNER' in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d['kind'] == 'port' and d['unit'] == 'iters/s' and d['value'] > 0 and d['cores'] >= 1


def test_multi_rank_preflight_fails_fast_and_readably():
    """`bench.py --gpus N` on a node with fewer than N visible GPUs (here: none) stops before any rank is launched, with a message that names the
    remedy -- not a torchrun traceback or a hang in the first collective."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(repo / 'bench.py'), '--gpus', '2', '--no-extras', '--no-cpu-baseline'], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'GPU(s) visible' in r.stderr and '--shared-device' in r.stderr and r.stdout.strip() == ''
