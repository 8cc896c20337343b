"""Loader + ctypes signatures of libfgs_hip.so (include/fgs_hip.h).

The product path has exactly one implementation: the gfx950 HIP library. If it is missing this module raises -- there is
no CPU or PyTorch fallback (the reference likewise refuses to run without its compiled `_C`,
FasterGSCudaBackend/__init__.py:13-20).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PACKAGE_ROOT = Path(__file__).resolve().parent.parent          # faster-gaussian-splatting_amd/
DEFAULT_LIBRARY = PACKAGE_ROOT / 'libfgs_hip.so'

FGS_BUF_PRIMITIVE, FGS_BUF_TILE, FGS_BUF_INSTANCE, FGS_BUF_BUCKET, FGS_BUF_SCRATCH = 0, 1, 2, 3, 4


class ExtensionError(ImportError):
    """Counterpart of Framework.ExtensionError raised by the reference's package init."""


class Settings(C.Structure):
    _fields_ = [
        ('w2c', C.c_void_p), ('cam_position', C.c_void_p), ('bg_color', C.c_void_p),
        ('active_sh_bases', C.c_int32), ('total_sh_bases_rest', C.c_int32), ('width', C.c_int32), ('height', C.c_int32),
        ('focal_x', C.c_float), ('focal_y', C.c_float), ('center_x', C.c_float), ('center_y', C.c_float),
        ('near_plane', C.c_float), ('far_plane', C.c_float), ('proper_antialiasing', C.c_int32),
    ]


class ForwardState(C.Structure):
    _fields_ = [('n_visible', C.c_int32), ('n_instances', C.c_int32), ('n_buckets', C.c_int32), ('selector', C.c_int32)]


class StageTime(C.Structure):
    _fields_ = [('name', C.c_char_p), ('total_ms', C.c_double), ('calls', C.c_int64)]


class BlobEntry(C.Structure):
    _fields_ = [('name', C.c_char_p), ('offset', C.c_size_t), ('bytes', C.c_size_t)]


SPLAT_RECORD_BYTES, ACC_RECORD_BYTES = 56, 36        # FGS_SPLAT_RECORD_BYTES / FGS_ACC_RECORD_BYTES

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t)

_P, _I32, _I64, _F64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double

_SIGNATURES = {
    'fgs_abi_version': (C.c_int32, []),
    'fgs_last_error': (C.c_char_p, []),
    'fgs_build_info': (C.c_char_p, []),
    'fgs_forward': (C.c_int32, [_P] * 6 + [_I32, C.POINTER(Settings), _P, RESIZE_FN, _P, C.POINTER(ForwardState), _P]),
    'fgs_forward_async': (C.c_int32, [_P] * 6 + [_I32, C.POINTER(Settings), _P, _I32, RESIZE_FN, _P, C.POINTER(ForwardState), _P]),
    'fgs_forward_counts': (C.c_int32, [_P, _I32, _P, _P]),
    'fgs_backward_scratch_bytes': (C.c_size_t, [_I32, _I32, _I32]),
    'fgs_backward': (C.c_int32, [_P] * 2 + [_P] * 5 + [_P] * 4 + [_P] * 6 + [_P, _P, _I32, C.POINTER(Settings), C.POINTER(ForwardState), _P]),
    'fgs_backward_live': (C.c_int32, [_P] * 2 + [_P] * 5 + [_P] * 4 + [_P] * 6 + [_P, _P, _I32, C.POINTER(Settings), C.POINTER(ForwardState), _P, _P]),
    'fgs_inference': (C.c_int32, [_P] * 6 + [_I32, C.POINTER(Settings), _P, _I32, _I32, RESIZE_FN, _P, C.POINTER(ForwardState), _P]),
    'fgs_pruning_scores': (C.c_int32, [_P] * 7 + [_I32, C.POINTER(Settings), RESIZE_FN, _P, C.POINTER(ForwardState), _P]),
    'fgs_adam_step': (C.c_int32, [_P] * 4 + [_I64, _I32, _F64, _F64, _F64, _F64, _P]),
    'fgs_adam_step_multi': (C.c_int32, [_I32] + [C.POINTER(_P)] * 4 + [C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_F64), _F64, _F64, _F64, _P]),
    'fgs_adam_step_multi_live': (C.c_int32, [_I32] + [C.POINTER(_P)] * 4 + [C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_F64), _F64, _F64, _F64, _P,
                                             C.POINTER(_I32), _P]),
    'fgs_backward_adam_fused': (C.c_int32, [_P] * 2 + [C.POINTER(_P)] * 3 + [_P] * 4 + [_P, _P, _I32, C.POINTER(Settings), C.POINTER(ForwardState),
                                            _I32, C.POINTER(_F64), _F64, _F64, _F64, _P]),
    'fgs_shard_preprocess': (C.c_int32, [_P] * 6 + [_I32, _I32, C.POINTER(Settings), _P, _P, RESIZE_FN, _P, _P]),
    'fgs_forward_from_records': (C.c_int32, [_P, _I32, _I32, C.POINTER(Settings), _P, RESIZE_FN, _P, C.POINTER(ForwardState), _P]),
    'fgs_backward_to_records': (C.c_int32, [_P] * 2 + [_P] * 4 + [_P, _P, _I32, C.POINTER(Settings), C.POINTER(ForwardState), _P]),
    'fgs_forward_from_shard_records': (C.c_int32, [_P, _I32, _I32, C.POINTER(_I32), _I32, C.POINTER(Settings), _P, RESIZE_FN, _P, C.POINTER(ForwardState), _P]),
    'fgs_backward_to_shard_records': (C.c_int32, [_P] * 2 + [_P] * 4 + [_P, _P, _I32, C.POINTER(_I32), _I32, C.POINTER(Settings), C.POINTER(ForwardState), _P]),
    'fgs_shard_backward_scratch_bytes': (C.c_size_t, [_I32, _I32]),
    'fgs_shard_backward': (C.c_int32, [_P, C.POINTER(_I32), _P] + [_P] * 5 + [_P] * 6 + [_P, _P, _I32, _I32, C.POINTER(Settings), _P]),
    'fgs_shard_backward_adam_fused': (C.c_int32, [_P, C.POINTER(_I32), _P] + [C.POINTER(_P)] * 3 + [_P, _P, _I32, _I32, C.POINTER(Settings), _I32,
                                                  C.POINTER(_F64), _F64, _F64, _F64, _P]),
    'fgs_blob_layout': (C.c_int32, [_I32] * 6 + [C.POINTER(BlobEntry), _I32]),
    'fgs_update_3d_filter': (C.c_int32, [_P] * 4 + [_I32, _I32, _I32] + [C.c_float] * 7 + [_P]),
    'fgs_relocation_table': (C.c_int32, [C.POINTER(C.c_float)]),
    'fgs_relocation_adjustment': (C.c_int32, [_P] * 6 + [_I32, _P]),
    'fgs_add_noise': (C.c_int32, [_P] * 5 + [_I32, C.c_float, _P]),
    'fgs_adc_scratch_bytes': (C.c_size_t, [_I32]),
    'fgs_adc_plan': (C.c_int32, [_P] * 4 + [_I32, C.c_float, C.c_float, _I32, C.c_float, C.c_float, _P, C.POINTER(_I32), _P]),
    'fgs_adc_apply': (C.c_int32, [C.POINTER(_P)] * 6 + [_P, _P, _I32, _I32, _P]),
    'fgs_gather_rows': (C.c_int32, [_I32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I32), _P, _I32, _P]),
    'fgs_morton_order_temp_bytes': (C.c_size_t, [_I32]),
    'fgs_morton_order': (C.c_int32, [_P, _P, _P, _P, _I32, _P, C.c_size_t, _P]),
    'fgs_l1_dssim_scratch_bytes': (C.c_size_t, [_I32, _I32]),
    'fgs_l1_dssim_loss': (C.c_int32, [_P, _P, _I32, _I32, C.c_float, C.c_float, _P, _P, _P, _P]),
    'fgs_l1_dssim_backward': (C.c_int32, [_P, _P, _I32, _I32, C.c_float, C.c_float, _P, _P, _P, _P]),
    'fgs_profile_enable': (C.c_int32, [_I32]),
    'fgs_profile_read': (C.c_int32, [C.POINTER(StageTime), _I32]),
    'fgs_debug_wave_selftest': (C.c_int32, [_P, _P]),
    'fgs_debug_radix_sort_temp_bytes': (C.c_size_t, [_I32, _I32]),
    'fgs_debug_radix_sort': (C.c_int32, [_P, _P, _P, _P, _I32, _I32, _I32, _P, C.c_size_t, _P]),
    'fgs_debug_depth_sort': (C.c_int32, [_P, _P, _P, _P, _I32, C.c_float, C.c_float, _P, C.c_size_t, _P]),
}
# libfgs_hip_dev.so only (-DFGS_DEV_SWITCHES): the A/B switchboard of tools/ and of the variant tests; bound when the library has them
_DEV_SIGNATURES = {
    'fgs_debug_set_backward_variant': (C.c_int32, [_I32]),
    'fgs_debug_set_option': (C.c_int32, [_I32, _I32]),
}

ABI_VERSION = 3                       # FGS_ABI_VERSION of include/fgs_hip.h
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
DEV_LIBRARY = PACKAGE_ROOT / 'libfgs_hip_dev.so'


def bind(path: os.PathLike | str) -> C.CDLL:
    """dlopen `path` and attach the argument/return types of every entry point declared in include/fgs_hip.h."""
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == the library does not export the declared ABI
        fn.restype, fn.argtypes = restype, argtypes
    for name, (restype, argtypes) in _DEV_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = restype, argtypes
    return lib


_LIB: C.CDLL | None = None


def library() -> C.CDLL:
    """The process-wide handle of libfgs_hip.so; raises ExtensionError if it was not built."""
    global _LIB
    if _LIB is None:
        path = Path(os.environ.get('FGS_HIP_LIBRARY', DEFAULT_LIBRARY))
        if not path.exists():
            raise ExtensionError(
                f'libfgs_hip.so not found at {path}. Build it with `make -C {PACKAGE_ROOT / "csrc"}` '
                f'(hipcc --offload-arch=gfx950) or `python __graft_entry__.py build`. There is no CPU fallback.')
        try:
            _LIB = bind(path)
        except OSError as exc:          # e.g. libamdhip64.so missing
            raise ExtensionError(f'failed to load {path}: {exc}') from exc
        if _LIB.fgs_abi_version() != ABI_VERSION:
            raise ExtensionError(f'{path} has ABI version {_LIB.fgs_abi_version()}, expected {ABI_VERSION}')
    return _LIB
