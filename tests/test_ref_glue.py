"""The reference-glue pin on the CPU: tests/golden/ref_glue_<scene>.npz / .trace.json were produced by the reference's own,
unmodified torch_bindings/*.py (loaded from /root/reference by path in the build container, tests/golden/make_ref_glue_golden.py) running on
this repository's `_C` bound to the CPU simulation of the product sources. Here, on the same simulation:

  * the package's own operators (FasterGSCudaBackend.diff_rasterize / rasterize / FusedAdam / ...), run through the same scenario, reproduce the
    fixture BIT FOR BIT -- rows a22 / a29 / a32: this package's autograd glue, optimizer and wrappers route their arguments as the reference's do;
  * the recorded `_C` call trace, replayed against `_C`, reproduces it bit for bit -- the fixture and the trace belong together;
  * (build container only) running the reference's files again gives the committed fixture: it is what the committed generator produces.
"""
import numpy as np
import pytest

import ref_glue


def _assert_identical(got: dict, want: dict, keys=None):
    keys = sorted(want) if keys is None else keys
    missing = [k for k in keys if k not in got]
    assert not missing, f'missing outputs: {missing}'
    for k in keys:
        a, b = np.asarray(got[k]), np.asarray(want[k])
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        assert a.tobytes() == b.tobytes(), f'{k}: differs from what the reference glue produced (max abs diff {np.abs(a.astype(np.float64) - b).max():.3e})'


@pytest.mark.parametrize('name', ref_glue.SCENES)
def test_package_operators_reproduce_the_reference_glue_bit_for_bit(name):
    want, _trace = ref_glue.load_fixture(name)
    with ref_glue.simulated_backend():
        got = ref_glue.run_scenario(ref_glue.package_ops(), name, 'cpu')
    assert sorted(got) == sorted(want)
    _assert_identical(got, want)
    # the scenario's edge cases really happened: a group without a gradient was skipped once, the others stepped three times (adam.py:15-25)
    assert int(want['it2_step_rotations']) == 2 and int(want['it2_step_means']) == 3
    assert np.array_equal(want['it1_param_rotations'], want['it0_param_rotations']) and not np.array_equal(want['it2_param_rotations'], want['it1_param_rotations'])
    # an empty densification_info leaves the statistics alone (api:136): two of three iterations counted
    assert want['densification_info'][0].max() == 2.0


@pytest.mark.parametrize('name', ref_glue.SCENES)
def test_recorded_call_trace_replays_bit_for_bit(name):
    want, trace = ref_glue.load_fixture(name)
    assert [c['fn'] for c in trace].count('adam_step') == 17 and sorted({c['fn'] for c in trace}) == sorted(ref_glue.C_ENTRY_POINTS)
    with ref_glue.simulated_backend():
        from FasterGSCudaBackend import _C
        got = ref_glue.replay_outputs(_C, name, trace, 'cpu')
    _assert_identical(got, want, [k for k in sorted(want) if '_step_' not in k])


def test_trace_states_the_reference_argument_routing():
    """What the trace pins, spelled out for one backward call: rasterization.py:56-104 of the reference (as executed)."""
    _want, trace = ref_glue.load_fixture('s0')
    fwd = next(i for i, c in enumerate(trace) if c['fn'] == 'forward' and trace[i + 1]['fn'] == 'backward')
    names = [a.get('t', a.get('v')) for a in trace[fwd + 1]['args']]
    assert names[:12] == ['densification_info', 'grad_image', f'c{fwd}.0', 'means', 'scales', 'rotations', 'opacities', 'sh_coefficients_rest',
                          f'c{fwd}.1', f'c{fwd}.2', f'c{fwd}.3', f'c{fwd}.4']
    assert names[12:15] == ['w2c', 'cam_position', 'bg_color'] and names[-3:] == [f'c{fwd}.5', f'c{fwd}.6', f'c{fwd}.7']
    adam = trace[fwd + 2]
    assert adam['fn'] == 'adam_step' and [a.get('t') for a in adam['args'][:4]] == [f'c{fwd + 1}.0', 'means', 'exp_avg:means', 'exp_avg_sq:means']


@pytest.mark.skipif(not ref_glue.reference_available(), reason='the reference tree exists in the build container only')
@pytest.mark.parametrize('name', ref_glue.SCENES)
def test_committed_fixture_is_what_the_reference_files_produce(name):
    import sys
    sys.path.insert(0, str(ref_glue.GOLDEN))
    import make_ref_glue_golden as gen
    want, trace = ref_glue.load_fixture(name)
    arrays, new_trace = gen.generate(name)
    assert new_trace == trace
    _assert_identical(arrays, want)
