"""Test infrastructure for the reference-glue pin: ONE scenario, written against the public operator names only, that
can be run (a) through the reference's own, unmodified torch_bindings/*.py loaded from /root/reference by path (build container only:
tests/golden/make_ref_glue_golden.py) and (b) through this package's operators, on the CPU simulation or on the MI355X. (a) produces the
committed fixtures tests/golden/ref_glue_<scene>.npz and the call trace ref_glue_<scene>.trace.json; (b) is compared with them.

What the scenario exercises, with the call shapes of the reference's callers (cited per block): Renderer.py:107-123 (`rasterize` under
inference_mode, CHW / HWC, clamped or not), Renderer.py:88-105 (`diff_rasterize` under no_grad on fresh tensors: scales + log(modifier), zeroed
sh_rest, empty densification_info), Renderer.py:141-156 (`update_pruning_scores`), Renderer.py:72-86 + Trainer.py:170-199 + Model.py:238-247
(three training iterations: keyword-argument `diff_rasterize` with a [2, N] and an empty(0) densification_info -> backward ->
`FusedAdam(param_groups, lr=0.0, eps=1e-15).step()`, one group without a gradient in the second iteration), Model.py:385-389
(`relocation_adjustment`), Model.py:476 (`add_noise`), Model.py:169-193 (`update_3d_filter`).

The recorder logs every call the glue makes into `FasterGSCudaBackend._C` (the eight entry points of bindings.cpp:12-21) as data: function name
and, per positional argument, either the scalar or the NAME of the tensor handed over (which input, which output of which earlier call). The
trace is therefore the reference's argument routing (save_for_backward order, as_tuple order, buffer_state order, Adam's state hand-over) as
executed, and `replay` re-issues it against any `_C` -- this is how the GPU suite checks rows a22 / a29 / a32 without the reference present.
"""
from __future__ import annotations

import contextlib
import importlib.util
import json
import math
import re
import sys
from pathlib import Path
from types import SimpleNamespace
from unittest import mock

import numpy as np
import torch

import helpers

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / 'tests' / 'golden'
REFERENCE_BINDINGS = Path('/root/reference/FasterGSCudaBackend/FasterGSCudaBackend/torch_bindings')
OP_NAMES = ('diff_rasterize', 'rasterize', 'update_pruning_scores', 'RasterizerSettings', 'FusedAdam', 'update_3d_filter',
            'relocation_adjustment', 'add_noise')
C_ENTRY_POINTS = ('forward', 'backward', 'inference', 'pruning_scores', 'adam_step', 'update_3d_filter', 'relocation_adjustment', 'add_noise')
# optimizer groups in the order and with the names of Model.py:238-245; learning rates of fastergs_garden.yaml (means: 1.6e-4 x extent)
GROUPS = (('means', 1.6e-4 * 4.8), ('sh_coefficients_0', 2.5e-3), ('sh_coefficients_rest', 1.25e-4), ('opacities', 2.5e-2), ('scales', 5e-3),
          ('rotations', 1e-3))
SCENES = ('s0', 'tiny_aa')


def scene(name: str):
    """(params, view, active_sh_bases, proper_antialiasing) of the two fixture scenes (the ones tests/golden/make_golden.py uses)."""
    from harness.scenes import View, make_s0
    if name == 's0':
        params, view = make_s0()
        return params, view, 16, False
    params, view = make_s0(seed=3, n=150)
    view = View(view.w2c, view.position, 48, 36, 40.0, 40.0, 24.0, 18.0, 0.2, 1e4, torch.tensor([0.2, 0.5, 0.7]))
    return params, view, 4, True


# ---- where the operators come from ----------------------------------------------------------------------------------------------------------
def package_ops() -> SimpleNamespace:
    import FasterGSCudaBackend as pkg
    return SimpleNamespace(**{n: getattr(pkg, n) for n in OP_NAMES})


def reference_ops() -> SimpleNamespace:
    """The reference's torch_bindings modules, executed from where they lie (never copied): they do `from FasterGSCudaBackend import _C`,
    which resolves to this repo's `_C`."""
    helpers.backend_modules()
    import FasterGSCudaBackend._C  # noqa: F401  (the module the reference files import)
    mods = {}
    for stem in ('rasterization', 'adam', 'densification', 'filter3d'):
        spec = importlib.util.spec_from_file_location(f'_reference_torch_bindings_{stem}', REFERENCE_BINDINGS / f'{stem}.py')
        mods[stem] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[stem])
    r, a, d, f = (mods[k] for k in ('rasterization', 'adam', 'densification', 'filter3d'))
    return SimpleNamespace(diff_rasterize=r.diff_rasterize, rasterize=r.rasterize, update_pruning_scores=r.update_pruning_scores,
                           RasterizerSettings=r.RasterizerSettings, FusedAdam=a.FusedAdam, update_3d_filter=f.update_3d_filter,
                           relocation_adjustment=d.relocation_adjustment, add_noise=d.add_noise)


@contextlib.contextmanager
def simulated_backend():
    """The package (its `_C` and its own operators) bound to the CPU simulation of the PRODUCT flavour of the sources for the duration of the block.
    Test infrastructure: the device guards of the public operators are lifted so that CPU tensors reach the simulation."""
    _lib, _backend = helpers.backend_modules()
    from FasterGSCudaBackend import aux_ops, rasterization
    saved = _backend._DEFAULT
    _backend._DEFAULT = helpers.sim_backend(product=True)
    rasterization.clear_live_blocks()
    try:
        with mock.patch.object(rasterization, '_require_gpu', lambda t: None), mock.patch.object(aux_ops, '_gpu', lambda t: None):
            yield _backend._DEFAULT
    finally:
        _backend._DEFAULT = saved
        rasterization.clear_live_blocks()


# ---- the recorder ---------------------------------------------------------------------------------------------------------------------------
_CALL_OUTPUT = re.compile(r'^c\d+\.\d+$')


def _key(t: torch.Tensor):
    return (t.data_ptr(), tuple(t.shape), str(t.dtype))


class Recorder:
    """Wraps the entry points of a `_C` module; `trace()` returns the calls as JSON-able data."""

    def __init__(self, c_module):
        self.c, self.calls, self.names, self.keep, self.saved, self.int_outputs = c_module, [], {}, [], {}, {}

    def note(self, name: str, t: torch.Tensor, alias_outputs: bool = False) -> None:
        """Names a tensor the scenario owns. alias_outputs: autograd may hand an output of `_C.backward` on as `.grad` or clone it first -- a
        tensor whose bytes equal an already named one of the same shape takes that name."""
        if t is None or t.numel() == 0:
            return
        self.keep.append(t)
        if alias_outputs and _key(t) not in self.names:
            for other in reversed(self.keep):          # the latest call output first
                k = _key(other)
                if _CALL_OUTPUT.match(self.names.get(k, '')) and other.shape == t.shape and other.dtype == t.dtype and other is not t \
                        and torch.equal(other, t):
                    self.names[_key(t)] = self.names[k]
                    return
        self.names.setdefault(_key(t), name)

    def __enter__(self):
        for fn in C_ENTRY_POINTS:
            self.saved[fn] = getattr(self.c, fn)
            setattr(self.c, fn, self._wrap(fn, self.saved[fn]))
        return self

    def __exit__(self, *exc):
        for fn, original in self.saved.items():
            setattr(self.c, fn, original)

    def _wrap(self, fn, original):
        def wrapper(*args, **kwargs):
            assert not kwargs, f'_C.{fn} is a pybind11 function of positional arguments (bindings.cpp:12-21)'
            index = len(self.calls)
            entry = {'fn': fn, 'args': [self._token(a) for a in args]}
            if fn == 'backward':          # n_instances, n_buckets, selector: integers the latest forward call returned, in whatever order the glue passes them
                latest = self.int_outputs
                for tok in entry['args']:
                    if tok.get('type') == 'int':
                        hits = [n for n, v in latest.items() if v == tok['v']]
                        if len(hits) == 1:
                            tok.clear()
                            tok['t'] = hits[0]
            self.calls.append(entry)
            out = original(*args)
            outs = out if isinstance(out, tuple) else (out,) if out is not None else ()
            entry['n_out'] = len(outs) if isinstance(out, tuple) else (1 if out is not None else 0)
            entry['tuple'] = isinstance(out, tuple)
            if fn == 'forward':
                self.int_outputs = {f'c{index}.{i}': o for i, o in enumerate(outs) if isinstance(o, int)}
            for i, o in enumerate(outs):
                if isinstance(o, torch.Tensor):
                    self.note(f'c{index}.{i}', o)
            return out
        return wrapper

    def _token(self, a):
        if isinstance(a, torch.Tensor):
            self.keep.append(a)
            if a.numel() == 0:
                return {'empty': list(a.shape), 'dtype': str(a.dtype)}
            return {'key': _key(a), 'tensor': a}
        if isinstance(a, (bool, int, float)):
            return {'v': a, 'type': type(a).__name__}
        raise TypeError(f'unexpected argument type {type(a)} in a _C call')

    def trace(self) -> list:
        """Resolves tensor tokens to names (late: Adam's moments are created inside the glue and named by the scenario afterwards)."""
        out = []
        for call in self.calls:
            args = []
            for tok in call['args']:
                if 'key' in tok:
                    name = self.names.get(tok['key'])
                    if name is None:       # e.g. the gradient autograd hands to backward: its bytes equal a named tensor (the upstream gradient)
                        for other in self.keep:
                            k = _key(other)
                            if k in self.names and other.shape == tok['tensor'].shape and other.dtype == tok['tensor'].dtype \
                                    and torch.equal(other, tok['tensor']):
                                name = self.names[k]
                                break
                    if name is None:
                        raise RuntimeError(f'_C.{call["fn"]}: a tensor argument of shape {tok["key"][1]} could not be attributed')
                    args.append({'t': name})
                else:
                    args.append(dict(tok))
            out.append({'fn': call['fn'], 'args': args, 'n_out': call['n_out'], 'tuple': call['tuple']})
        return out


def replay(c_module, trace: list, env: dict, device, start: int = 0, stop: int | None = None) -> dict:
    """Re-issues a recorded call sequence against `c_module` with the tensors of `env` (name -> tensor on `device`). Outputs are stored under
    their call names; Adam moments (`exp_avg:<group>`, `exp_avg_sq:<group>`) are created as zeros on first use, as adam.py:17-21 does."""
    for index in range(start, len(trace) if stop is None else stop):
        call = trace[index]
        args = []
        for tok in call['args']:
            if 't' in tok:
                name = tok['t']
                if name not in env and name.startswith('exp_avg'):
                    env[name] = torch.zeros_like(env[name.split(':', 1)[1]])
                args.append(env[name])
            elif 'empty' in tok:
                args.append(torch.empty(tok['empty'], dtype=getattr(torch, tok['dtype'].split('.')[1]), device=device))
            else:
                args.append({'bool': bool, 'int': int, 'float': float}[tok['type']](tok['v']))
        with torch.no_grad():
            out = getattr(c_module, call['fn'])(*args)
        outs = out if isinstance(out, tuple) else (out,) if out is not None else ()
        assert isinstance(out, tuple) == call['tuple'] and len(outs) == call['n_out'], f'_C.{call["fn"]} returned a different structure'
        for i, o in enumerate(outs):
            env[f'c{index}.{i}'] = o
    return env


# ---- the scenario ---------------------------------------------------------------------------------------------------------------------------
def aux_inputs(seed: int = 11, n: int = 400) -> dict:
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    nrm = lambda *s: torch.randn(*s, generator=g)
    w2c = torch.eye(4)
    w2c[2, 3] = 3.0
    return {
        'reloc_opacities': 0.05 + 0.9 * u(n), 'reloc_scales': 0.01 + 0.2 * u(n, 3), 'reloc_counts': torch.randint(1, 9, (n,), generator=g),
        'noise_raw_scales': 0.3 * nrm(n, 3) - 3.0, 'noise_raw_rotations': nrm(n, 4), 'noise_raw_opacities': nrm(n, 1),
        'noise_means': u(n, 3) * 2.0 - 1.0, 'noise_samples': nrm(n, 3),
        'f3d_positions': (u(n, 3) * 2.0 - 1.0) * 1.5, 'f3d_w2c': w2c,
        'f3d_filter': torch.full((n, 1), torch.finfo(torch.float32).max), 'f3d_mask': torch.zeros((n, 1), dtype=torch.bool),
    }


def run_scenario(ops, name: str, device, recorder: Recorder | None = None) -> dict:
    """Runs the scenario of the module docstring with the operators in `ops`; returns {key: numpy array} (the fixture's content)."""
    note = recorder.note if recorder is not None else (lambda *a, **k: None)
    params, view, K, aa = scene(name)
    n = params['means'].shape[0]
    out: dict = {}
    cpu = lambda t: t.detach().cpu().numpy().copy()
    P = {k: torch.nn.Parameter(params[k].clone().to(device).contiguous()) for k in helpers.NAMES}
    for k, p in P.items():
        note(k, p)
    w2c, pos, bg = view.w2c.to(device), view.position.to(device), view.background_color.to(device)
    note('w2c', w2c), note('cam_position', pos), note('bg_color', bg)
    RS = ops.RasterizerSettings(w2c, pos, bg, K, view.width, view.height, view.focal_x, view.focal_y, view.center_x, view.center_y,
                                view.near_plane, view.far_plane, aa)
    six = lambda d: dict(means=d['means'], scales=d['scales'], rotations=d['rotations'], opacities=d['opacities'],
                         sh_coefficients_0=d['sh_coefficients_0'], sh_coefficients_rest=d['sh_coefficients_rest'])

    # Renderer.py:107-123 -- the benchmark / inference path
    with torch.inference_mode():
        for to_chw in (True, False):
            for clamp in (True, False):
                out[f'rasterize_chw{int(to_chw)}_clamp{int(clamp)}'] = cpu(ops.rasterize(**six(P), rasterizer_settings=RS, to_chw=to_chw, clamp_output=clamp))
    # Renderer.py:88-105 -- render_image_inference: the training operator under no_grad on fresh, non-parameter tensors
    with torch.no_grad():
        mod = dict(six(P), scales=P['scales'] + math.log(0.7), sh_coefficients_rest=torch.zeros_like(P['sh_coefficients_rest']))
        note('scales_modified', mod['scales']), note('sh_rest_zeroed', mod['sh_coefficients_rest'])
        image = ops.diff_rasterize(**mod, densification_info=torch.empty(0), rasterizer_settings=RS)
        out['no_grad_image'] = cpu(image.clamp(0.0, 1.0).permute(1, 2, 0))
    # Renderer.py:141-156 -- compute_pruning_scores
    with torch.inference_mode():
        scores = torch.zeros(n, device=device, dtype=torch.float32)
        note('scores', scores)
        assert ops.update_pruning_scores(scores=scores, **six(P), rasterizer_settings=RS) is None
        out['pruning_scores'] = cpu(scores)

    # Model.py:238-247 + Trainer.py:170-199 -- three training iterations
    opt = ops.FusedAdam([{'params': [P[g]], 'lr': lr, 'name': g} for g, lr in GROUPS], lr=0.0, eps=1e-15)
    dens = torch.zeros(2, n, device=device)
    gi = torch.randn((3, view.height, view.width), generator=torch.Generator().manual_seed(5)).to(device)
    note('densification_info', dens), note('grad_image', gi)
    for it in range(3):
        for group in opt.param_groups:          # Model.py:255-260: the means' learning rate follows a schedule
            if group['name'] == 'means':
                group['lr'] = GROUPS[0][1] * (0.9 ** it)
        image = ops.diff_rasterize(**six(P), densification_info=dens if it != 1 else torch.empty(0), rasterizer_settings=RS)
        (image * gi).sum().backward()
        out[f'it{it}_image'] = cpu(image)
        for k in helpers.NAMES:
            note(f'grad{it}:{k}', P[k].grad, alias_outputs=True)
            out[f'it{it}_grad_{k}'] = cpu(P[k].grad)
        if it == 1:
            P['rotations'].grad = None          # adam.py:15-16: a group without a gradient is skipped (no state change, no step count)
        opt.step()
        for k in helpers.NAMES:
            state = opt.state.get(P[k], {})
            out[f'it{it}_param_{k}'] = cpu(P[k])
            out[f'it{it}_step_{k}'] = np.asarray(int(state.get('step', 0)))
            if 'exp_avg' in state:
                note(f'exp_avg:{k}', state['exp_avg']), note(f'exp_avg_sq:{k}', state['exp_avg_sq'])
                out[f'it{it}_exp_avg_{k}'], out[f'it{it}_exp_avg_sq_{k}'] = cpu(state['exp_avg']), cpu(state['exp_avg_sq'])
        opt.zero_grad(set_to_none=True)
    out['densification_info'] = cpu(dens)

    # Model.py:385-389, 476, 169-193 -- the remaining operators, on their own inputs
    aux = {k: v.to(device) for k, v in aux_inputs().items()}
    for k, v in aux.items():
        note(k, v)
    new_op, new_sc = ops.relocation_adjustment(aux['reloc_opacities'], aux['reloc_scales'], aux['reloc_counts'])
    out['reloc_new_opacities'], out['reloc_new_scales'] = cpu(new_op), cpu(new_sc)
    with mock.patch('torch.randn_like', lambda t, *a, **k: aux['noise_samples']):          # densification.py:20 draws the samples in the wrapper
        assert ops.add_noise(aux['noise_raw_scales'], aux['noise_raw_rotations'], aux['noise_raw_opacities'], aux['noise_means'], 5e5 * 1.6e-5) is None
    out['noise_means_after'] = cpu(aux['noise_means'])
    assert ops.update_3d_filter(aux['f3d_positions'], aux['f3d_w2c'], aux['f3d_filter'], aux['f3d_mask'], 64, 48, 60.0, 60.0, 32.0, 24.0, 0.2,
                                0.15, math.sqrt(0.2) / 60.0) is None
    out['f3d_filter_after'], out['f3d_mask_after'] = cpu(aux['f3d_filter']), cpu(aux['f3d_mask'])
    return out


def replay_environment(name: str, device) -> dict:
    """The named tensors a trace of `run_scenario(name)` refers to, as the scenario creates them."""
    params, view, _K, _aa = scene(name)
    n = params['means'].shape[0]
    env = {k: params[k].clone().to(device).contiguous() for k in helpers.NAMES}
    env.update(w2c=view.w2c.to(device), cam_position=view.position.to(device), bg_color=view.background_color.to(device),
               scales_modified=env['scales'] + math.log(0.7), sh_rest_zeroed=torch.zeros_like(env['sh_coefficients_rest']),
               scores=torch.zeros(n, device=device), densification_info=torch.zeros(2, n, device=device),
               grad_image=torch.randn((3, view.height, view.width), generator=torch.Generator().manual_seed(5)).to(device))
    env.update({k: v.to(device) for k, v in aux_inputs().items()})
    return env


def replay_outputs(c_module, name: str, trace: list, device) -> dict:
    """Replays `trace` call by call against `c_module` and collects what the scenario collects, under the fixture's keys (step counts excepted:
    they are arguments of the recorded calls)."""
    env = replay_environment(name, device)
    cpu = lambda t: t.detach().cpu().numpy().copy()
    out, iteration, n_forward = {}, -1, 0

    def snapshot_optimizer(it):
        for k in helpers.NAMES:
            out[f'it{it}_param_{k}'] = cpu(env[k])
            if f'exp_avg:{k}' in env:
                out[f'it{it}_exp_avg_{k}'], out[f'it{it}_exp_avg_sq_{k}'] = cpu(env[f'exp_avg:{k}']), cpu(env[f'exp_avg_sq:{k}'])

    for index, call in enumerate(trace):
        fn = call['fn']
        if fn != 'adam_step' and index > 0 and trace[index - 1]['fn'] == 'adam_step':
            snapshot_optimizer(iteration)
        replay(c_module, trace, env, device, index, index + 1)
        if fn == 'inference':
            to_chw, clamp = call['args'][-2]['v'], call['args'][-1]['v']
            out[f'rasterize_chw{int(to_chw)}_clamp{int(clamp)}'] = cpu(env[f'c{index}.0'])
        elif fn == 'forward':
            if n_forward == 0:
                out['no_grad_image'] = cpu(env[f'c{index}.0'].clamp(0.0, 1.0).permute(1, 2, 0))
            else:
                iteration += 1
                out[f'it{iteration}_image'] = cpu(env[f'c{index}.0'])
            n_forward += 1
        elif fn == 'backward':
            for i, k in enumerate(helpers.NAMES):
                out[f'it{iteration}_grad_{k}'] = cpu(env[f'c{index}.{i}'])
        elif fn == 'pruning_scores':
            out['pruning_scores'] = cpu(env['scores'])
        elif fn == 'relocation_adjustment':
            out['reloc_new_opacities'], out['reloc_new_scales'] = cpu(env[f'c{index}.0']), cpu(env[f'c{index}.1'])
        elif fn == 'add_noise':
            out['noise_means_after'] = cpu(env['noise_means'])
        elif fn == 'update_3d_filter':
            out['f3d_filter_after'], out['f3d_mask_after'] = cpu(env['f3d_filter']), cpu(env['f3d_mask'])
    out['densification_info'] = cpu(env['densification_info'])
    return out


def load_fixture(name: str):
    data = dict(np.load(GOLDEN / f'ref_glue_{name}.npz'))
    trace = json.loads((GOLDEN / f'ref_glue_{name}.trace.json').read_text())
    return data, trace


def reference_available() -> bool:
    return (REFERENCE_BINDINGS / 'rasterization.py').exists()


if str(REPO / 'tests') not in sys.path:
    sys.path.insert(0, str(REPO / 'tests'))
