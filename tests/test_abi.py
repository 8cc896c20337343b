"""CPU tests of the drop-in boundary: libfgs_hip.so loads, exports every symbol include/fgs_hip.h declares, validates
arguments without touching a GPU, and the package refuses to import without the library (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

import helpers

REPO = Path(__file__).resolve().parent.parent
LIB = REPO / 'faster-gaussian-splatting_amd' / 'libfgs_hip.so'


@pytest.fixture(scope='module')
def hip_lib():
    if not LIB.exists():
        subprocess.run(['make', '-C', str(LIB.parent / 'csrc'), '-j8'], check=True)
    _lib, _ = helpers.backend_modules()
    return _lib.bind(LIB)


def test_header_symbols_are_exported_and_bound(hip_lib):
    header = (REPO / 'include' / 'fgs_hip.h').read_text()
    dev_block = re.search(r'#ifdef FGS_DEV_SWITCHES\n(.*?)#endif /\* FGS_DEV_SWITCHES \*/', header, re.S)
    product_header = header.replace(dev_block.group(0), '')
    find = lambda text: set(re.findall(r'^(?:int32_t|size_t|const char\*)\s+(fgs_[a-z0-9_]+)\s*\(', text, re.M))
    declared, dev_declared = find(product_header), find(dev_block.group(1))
    _lib, _ = helpers.backend_modules()
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert dev_declared == set(_lib._DEV_SIGNATURES) == {'fgs_debug_set_backward_variant', 'fgs_debug_set_option'}
    for name in declared:
        assert getattr(hip_lib, name) is not None
    # the product library has no process-wide switches; the dev library (same sources, -DFGS_DEV_SWITCHES) has everything
    for name in dev_declared:
        assert not hasattr(hip_lib, name), name
    dev = _lib.DEV_LIBRARY
    if not dev.exists():
        subprocess.run(['make', '-C', str(dev.parent / 'csrc'), '-j8', 'dev'], check=True)
    dev_lib = _lib.bind(dev)
    for name in declared | dev_declared:
        assert getattr(dev_lib, name) is not None
    assert b'dev-switches' in dev_lib.fgs_build_info() and b'dev' not in hip_lib.fgs_build_info()
    assert hip_lib.fgs_abi_version() == 3          # FGS_ABI_VERSION (include/fgs_hip.h)
    assert b'gfx950' in hip_lib.fgs_build_info()
    # ... and no switch variables either: every A/B switch of the sources (FGS_SWITCH, csrc/fgs_kernels.h) is a compile-time constant in the product
    data = lambda lib: {ln.split()[-1] for ln in subprocess.run(['nm', '-C', str(lib)], check=True, capture_output=True, text=True).stdout.splitlines()
                        if re.search(r' [bBdDuV] (fgs::|\(anonymous namespace\)::)g_', ln)}
    assert data(LIB) <= {'(anonymous', 'namespace)::g_prof', 'namespace)::g_error'}, data(LIB)      # fgs_profile_enable's recorder, fgs_last_error's buffer
    assert {'fgs::g_backward_variant', 'fgs::g_depth_sort_mode', 'fgs::g_tile_row_group'} <= data(dev)


def test_argument_validation_without_gpu(hip_lib):
    _lib, _ = helpers.backend_modules()
    st = _lib.ForwardState()
    cb = _lib.RESIZE_FN(lambda u, w, n: 0)
    assert hip_lib.fgs_forward(None, None, None, None, None, None, 0, None, None, cb, None, C.byref(st), None) == -1
    assert b'settings' in hip_lib.fgs_last_error()
    S = _lib.Settings(1, 1, 1, 16, 15, 0, 128, 1.0, 1.0, 0.0, 0.0, 0.2, 100.0, 0)
    assert hip_lib.fgs_forward(None, None, None, None, None, None, 0, C.byref(S), 1, cb, None, C.byref(st), None) == -1
    assert b'image size' in hip_lib.fgs_last_error()
    assert hip_lib.fgs_adam_step_multi(9, None, None, None, None, None, None, None, 0.9, 0.999, 1e-15, None) == -1
    assert hip_lib.fgs_backward_scratch_bytes(1000, 128, 128) > 1000 * 48
    assert hip_lib.fgs_backward_scratch_bytes(-1, 128, 128) == 0
    entries = (_lib.BlobEntry * 16)()
    k = hip_lib.fgs_blob_layout(0, 1000, 128, 128, 0, 0, entries, 16)
    names = [entries[i].name.decode() for i in range(k)]
    assert names[:2] == ['rec', 'n_touched'] and entries[0].bytes == 48000 and all(entries[i].offset % 256 == 0 for i in range(k))


def test_package_fails_loudly_without_the_library(tmp_path):
    code = ("import sys; sys.path.insert(0, r'%s'); import FasterGSCudaBackend" % (REPO / 'faster-gaussian-splatting_amd'))
    env = dict(os.environ, FGS_HIP_LIBRARY=str(tmp_path / 'missing.so'))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'libfgs_hip.so not found' in r.stderr and 'no CPU fallback' in r.stderr


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import, link or load it."""
    for path in (REPO / 'faster-gaussian-splatting_amd').rglob('*'):
        if path.suffix in {'.py', '.hip', '.h'} or path.name == 'Makefile':
            text = path.read_text()
            assert 'oracle' not in text.lower() or path.name in {'fgs_math.h', 'preprocess.hip', 'binning.hip', 'Makefile'}, path
            assert 'libfgs_oracle' not in text and 'import oracle' not in text and 'from oracle' not in text, path


def test_cpu_tensors_are_rejected_by_the_public_operators(hip_lib):
    import torch
    import FasterGSCudaBackend as B
    from harness.scenes import make_s0
    params, view = make_s0(n=8)
    _, RS = helpers.settings_pair(view)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        B.diff_rasterize(*[params[k] for k in helpers.NAMES], torch.empty(0), RS)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        B.rasterize(*[params[k] for k in helpers.NAMES], RS, True)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        B.add_noise(params['scales'], params['rotations'], params['opacities'], params['means'], 1e-3)


def test_c_module_has_the_eight_reference_entry_points():
    """`FasterGSCudaBackend._C` mirrors the pybind11 module of the reference (torch_bindings/bindings.cpp:12-21): same eight names,
    same positional parameter lists (rasterization_api.h:8-106, adam.h:7-16, filter3d.h:7-20, densification_api.h:8-21)."""
    import inspect
    from FasterGSCudaBackend import _C
    common = ['w2c', 'cam_position', 'bg_color', 'active_sh_bases', 'width', 'height', 'focal_x', 'focal_y', 'center_x', 'center_y',
              'near_plane', 'far_plane', 'proper_antialiasing']
    six = ['means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest']
    expected = {
        'forward': six + common,
        'backward': ['densification_info', 'grad_image', 'image', 'means', 'scales', 'rotations', 'opacities', 'sh_coefficients_rest',
                     'primitive_buffers', 'tile_buffers', 'instance_buffers', 'bucket_buffers'] + common +
                    ['n_instances', 'n_buckets', 'instance_primitive_indices_selector'],
        'inference': six + common + ['to_chw', 'clamp_output'],
        'pruning_scores': ['scores'] + six + common,
        'adam_step': ['param_grad', 'param', 'exp_avg', 'exp_avg_sq', 'step_count', 'learning_rate', 'beta1', 'beta2', 'epsilon'],
        'update_3d_filter': ['positions', 'w2c', 'filter_3d', 'visibility_mask', 'width', 'height', 'focal_x', 'focal_y', 'center_x',
                             'center_y', 'near_plane', 'clipping_tolerance', 'distance2filter'],
        'relocation_adjustment': ['old_opacities', 'old_scales', 'n_samples_per_primitive'],
        'add_noise': ['raw_scales', 'raw_rotations', 'raw_opacities', 'random_samples', 'means', 'current_lr'],
    }
    public = {n for n, f in vars(_C).items() if inspect.isfunction(f) and not n.startswith('_')}
    assert public == set(expected), public ^ set(expected)
    for name, params in expected.items():
        assert list(inspect.signature(getattr(_C, name)).parameters) == params, name
