"""ctypes front-end of the CPU oracle (oracle/fgs_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see fgs_oracle.c header): the reference has no golden vectors and cannot run here.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does.

The functions chain the oracle stages in the order of the reference's host code:
  forward   = rasterization/src/forward.cu:11-259   (K1 .. K10)
  inference = rasterization/src/inference.cu:11-226
  backward  = rasterization/src/backward.cu:8-125 + rasterization_api.cu:127-134 (zero-filled grads)
  adam_step = adam/src/adam.cu:36-71
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

TILE_W, TILE_H, BLOCK_BLEND = 16, 12, 192


class _Settings(C.Structure):
    _fields_ = [
        ('w2c', C.c_float * 12), ('cam_pos', C.c_float * 3), ('bg', C.c_float * 3),
        ('active_sh_bases', C.c_int), ('total_sh_rest', C.c_int), ('width', C.c_int), ('height', C.c_int),
        ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float),
        ('near_plane', C.c_float), ('far_plane', C.c_float), ('proper_aa', C.c_int),
    ]


@dataclass
class Settings:
    """Numpy mirror of RasterizerSettings (torch_bindings/rasterization.py:8-38)."""
    w2c: np.ndarray            # [>=3, 4] row-major world-to-camera
    cam_position: np.ndarray   # [3]
    bg_color: np.ndarray       # [3]
    active_sh_bases: int
    width: int
    height: int
    focal_x: float
    focal_y: float
    center_x: float
    center_y: float
    near_plane: float
    far_plane: float
    proper_antialiasing: bool = False

    def to_c(self, total_sh_rest: int) -> _Settings:
        s = _Settings()
        w = np.ascontiguousarray(self.w2c, dtype=np.float32).reshape(-1)[:12]
        s.w2c[:] = w.tolist()
        s.cam_pos[:] = np.asarray(self.cam_position, dtype=np.float32).reshape(-1).tolist()
        s.bg[:] = np.asarray(self.bg_color, dtype=np.float32).reshape(-1).tolist()
        s.active_sh_bases, s.total_sh_rest = int(self.active_sh_bases), int(total_sh_rest)
        s.width, s.height = int(self.width), int(self.height)
        s.fx, s.fy, s.cx, s.cy = self.focal_x, self.focal_y, self.center_x, self.center_y
        s.near_plane, s.far_plane = self.near_plane, self.far_plane
        s.proper_aa = int(bool(self.proper_antialiasing))
        return s


def build(force: bool = False) -> Path:
    so = _HERE / 'libfgs_oracle.so'
    so64 = _HERE / 'libfgs_oracle64.so'
    src = _HERE / 'fgs_oracle.c'
    if force or not so.exists() or not so64.exists() or min(so.stat().st_mtime, so64.stat().st_mtime) < src.stat().st_mtime:
        subprocess.run(['make', '-C', str(_HERE), '-B', 'all'], check=True, capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.orc_preprocess.restype = C.c_int
        _LIB.orc_ranges_and_buckets.restype = C.c_uint
        _LIB.orc_extract_end_bit.restype = C.c_int
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def num_threads() -> int:
    return int(lib().orc_num_threads())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def grid_of(width: int, height: int):
    gw, gh = (width + TILE_W - 1) // TILE_W, (height + TILE_H - 1) // TILE_H
    return gw, gh, gw * gh


def end_bit_of(n_tiles: int) -> int:
    return int(lib().orc_extract_end_bit(C.c_uint(n_tiles - 1)))


def forward(means, scales, rotations, opacities, sh0, sh_rest, settings: Settings, *, bucket_size: int = 32,
            inference: bool = False, to_chw: bool = True, clamp_output: bool = True) -> dict:
    """Runs K1..K10 and returns every intermediate (names follow buffer_utils.h:45-163)."""
    L = lib()
    means, scales, rotations = _f32(means).reshape(-1, 3), _f32(scales).reshape(-1, 3), _f32(rotations).reshape(-1, 4)
    opacities, sh0 = _f32(opacities).reshape(-1), _f32(sh0).reshape(-1, 3)
    N = means.shape[0]
    sh_rest = _f32(sh_rest)
    total_rest = sh_rest.shape[1] if sh_rest.ndim == 3 else (sh_rest.size // (3 * N) if N else 0)
    sh_rest = sh_rest.reshape(N, total_rest, 3)
    S = settings.to_c(total_rest)
    W, H = settings.width, settings.height
    gw, gh, T = grid_of(W, H)
    out = {'N': N, 'grid': (gw, gh), 'T': T, 'end_bit': end_bit_of(T), 'bucket_size': bucket_size}

    n_touched = np.zeros(N, np.uint32)
    screen_bounds = np.zeros((N, 4), np.uint16)
    mean2d = np.zeros((N, 2), np.float32)
    conic_opacity = np.zeros((N, 4), np.float32)
    color = np.zeros((N, 3), np.float32)
    depth_keys = np.zeros(max(N, 1), np.uint32)
    prim_idx = np.zeros(max(N, 1), np.uint32)
    n_inst = C.c_uint(0)
    V = L.orc_preprocess(N, _p(means), _p(scales), _p(rotations), _p(opacities), _p(sh0), _p(sh_rest), C.byref(S),
                         int(inference), _p(n_touched), _p(screen_bounds), _p(mean2d), _p(conic_opacity), _p(color),
                         _p(depth_keys), _p(prim_idx), C.byref(n_inst))
    I = int(n_inst.value)
    depth_keys, prim_idx = depth_keys[:V].copy(), prim_idx[:V].copy()
    out.update(V=V, I=I, n_touched=n_touched, screen_bounds=screen_bounds, mean2d=mean2d, conic_opacity=conic_opacity,
               color=color, depth_keys_unsorted=depth_keys.copy(), prim_idx_unsorted=prim_idx.copy())
    L.orc_sort_pairs(V, _p(depth_keys), _p(prim_idx), 32)
    out.update(depth_keys=depth_keys, prim_idx=prim_idx)

    offsets = np.zeros(max(V, 1), np.uint32)
    inst_keys = np.zeros(max(I, 1), np.uint32)
    inst_prims = np.zeros(max(I, 1), np.uint32)
    L.orc_create_instances(V, _p(prim_idx), _p(n_touched), _p(screen_bounds), _p(mean2d), _p(conic_opacity), gw,
                           _p(offsets), _p(inst_keys), _p(inst_prims))
    out.update(offsets=offsets[:V], inst_keys_unsorted=inst_keys[:I].copy(), inst_prims_unsorted=inst_prims[:I].copy())
    L.orc_sort_pairs(I, _p(inst_keys), _p(inst_prims), out['end_bit'])
    inst_keys, inst_prims = inst_keys[:I], inst_prims[:I]
    out.update(inst_keys=inst_keys, inst_prims=inst_prims)

    ranges = np.zeros((T, 2), np.uint32)
    n_buckets = np.zeros(T, np.uint32)
    bucket_offsets = np.zeros(T, np.uint32)
    B = int(L.orc_ranges_and_buckets(I, _p(np.ascontiguousarray(inst_keys)), T, bucket_size, _p(ranges), _p(n_buckets),
                                     _p(bucket_offsets)))
    out.update(ranges=ranges, n_buckets=n_buckets, bucket_offsets=bucket_offsets, B=B)

    P = W * H
    inst_prims_c = np.ascontiguousarray(inst_prims) if I > 0 else np.zeros(1, np.uint32)
    if inference:
        image = np.zeros((3, H, W) if to_chw else (H, W, 3), np.float32)
        L.orc_blend_forward(1, int(to_chw), int(clamp_output), bucket_size, _p(ranges), _p(bucket_offsets), _p(inst_prims_c),
                            _p(screen_bounds), _p(mean2d), _p(conic_opacity), _p(color), C.byref(S), _p(image),
                            None, None, None, None, None)
        out['image'] = image
        return out
    image = np.zeros((3, H, W), np.float32)
    final_T = np.ones(P, np.float32)
    n_processed = np.zeros(P, np.uint32)
    max_n_processed = np.zeros(T, np.uint32)
    bucket_tile_index = np.zeros(max(B, 1), np.uint32)
    bucket_ckpt = np.zeros((max(B, 1), BLOCK_BLEND, 4), np.float32)
    L.orc_blend_forward(0, 1, 0, bucket_size, _p(ranges), _p(bucket_offsets), _p(inst_prims_c), _p(screen_bounds), _p(mean2d),
                        _p(conic_opacity), _p(color), C.byref(S), _p(image), _p(final_T), _p(n_processed),
                        _p(max_n_processed), _p(bucket_tile_index), _p(bucket_ckpt))
    out.update(image=image, final_T=final_T, n_processed=n_processed, max_n_processed=max_n_processed,
               bucket_tile_index=bucket_tile_index[:B], bucket_ckpt=bucket_ckpt[:B], _S=S,
               _inputs=(means, scales, rotations, opacities, sh0, sh_rest))
    return out


def reblend(fwd: dict, settings: Settings, mean2d, conic_opacity, color) -> dict:
    """Test support: K10 of `fwd` again on OTHER per-Gaussian records (the device's own, say) over the same discrete structure (instance lists, ranges,
    buckets, screen bounds). Returns a copy of `fwd` with the records and every blend output replaced -- what `backward` and `threshold_risk`
    need to follow. Isolates the two blend kernels from K1: a needle-shaped Gaussian's conic is cov / det with a cancelling det, and one ulp of
    exp moves it by up to 2e-3 (profiles/r06_fuzz_sweeps.txt); with the same records on both sides only the blend arithmetic is compared."""
    L = lib()
    out = dict(fwd)
    out['mean2d'] = np.ascontiguousarray(_f32(mean2d).reshape(-1, 2)); out['conic_opacity'] = np.ascontiguousarray(_f32(conic_opacity).reshape(-1, 4))
    out['color'] = np.ascontiguousarray(_f32(color).reshape(-1, 3))
    W, H, T, B, I = settings.width, settings.height, fwd['T'], fwd['B'], fwd['I']
    inst_prims_c = np.ascontiguousarray(fwd['inst_prims']) if I > 0 else np.zeros(1, np.uint32)
    image = np.zeros((3, H, W), np.float32)
    final_T = np.ones(W * H, np.float32)
    n_processed = np.zeros(W * H, np.uint32)
    max_n_processed = np.zeros(T, np.uint32)
    bucket_tile_index = np.zeros(max(B, 1), np.uint32)
    bucket_ckpt = np.zeros((max(B, 1), BLOCK_BLEND, 4), np.float32)
    L.orc_blend_forward(0, 1, 0, fwd['bucket_size'], _p(fwd['ranges']), _p(fwd['bucket_offsets']), _p(inst_prims_c), _p(fwd['screen_bounds']),
                        _p(out['mean2d']), _p(out['conic_opacity']), _p(out['color']), C.byref(fwd['_S']), _p(image), _p(final_T), _p(n_processed),
                        _p(max_n_processed), _p(bucket_tile_index), _p(bucket_ckpt))
    out.update(image=image, final_T=final_T, n_processed=n_processed, max_n_processed=max_n_processed,
               bucket_tile_index=bucket_tile_index[:B], bucket_ckpt=bucket_ckpt[:B])
    return out


def backward(fwd: dict, settings: Settings, grad_image, densification_info: np.ndarray | None = None) -> dict:
    """K11 + K12 on the state returned by forward(); grads are zero-initialised as in rasterization_api.cu:127-134."""
    L = lib()
    means, scales, rotations, opacities, sh0, sh_rest = fwd['_inputs']
    N, B, S = fwd['N'], fwd['B'], fwd['_S']
    grad_image = _f32(grad_image).reshape(3, settings.height, settings.width)
    g = {
        'means': np.zeros((N, 3), np.float32), 'scales': np.zeros((N, 3), np.float32),
        'rotations': np.zeros((N, 4), np.float32), 'opacities': np.zeros((N, 1), np.float32),
        'sh0': np.zeros((N, 1, 3), np.float32), 'sh_rest': np.zeros(sh_rest.shape, np.float32),
    }
    grad_mean2d = np.zeros((N, 2), np.float32)
    grad_conic = np.zeros((3, N), np.float32)
    inst_prims = np.ascontiguousarray(fwd['inst_prims']) if fwd['I'] > 0 else np.zeros(1, np.uint32)
    bti = np.ascontiguousarray(fwd['bucket_tile_index']) if B > 0 else np.zeros(1, np.uint32)
    ckpt = np.ascontiguousarray(fwd['bucket_ckpt']) if B > 0 else np.zeros((1, BLOCK_BLEND, 4), np.float32)
    L.orc_blend_backward(N, B, fwd['bucket_size'], _p(fwd['ranges']), _p(fwd['bucket_offsets']), _p(inst_prims),
                         _p(fwd['mean2d']), _p(fwd['conic_opacity']), _p(fwd['color']), C.byref(S), _p(grad_image),
                         _p(fwd['image']), _p(fwd['final_T']), _p(fwd['max_n_processed']), _p(fwd['n_processed']),
                         _p(bti), _p(ckpt), _p(grad_mean2d), _p(grad_conic), _p(g['opacities']), _p(g['sh0']))
    g['_grad_mean2d'] = grad_mean2d.copy()
    g['_grad_conic'] = grad_conic.copy()
    g['_grad_opacity_acc'] = g['opacities'].copy()
    g['_grad_color_acc'] = g['sh0'].copy()
    dens = None
    if densification_info is not None and densification_info.size > 0:
        assert densification_info.dtype == np.float32 and densification_info.flags['C_CONTIGUOUS']
        dens = _p(densification_info)
    L.orc_preprocess_backward(N, _p(means), _p(scales), _p(rotations), _p(opacities), _p(sh_rest), C.byref(S),
                              _p(fwd['n_touched']), _p(grad_mean2d), _p(grad_conic), _p(g['means']), _p(g['scales']),
                              _p(g['rotations']), _p(g['opacities']), _p(g['sh0']), _p(g['sh_rest']), dens)
    return g


# ---- double-precision evaluation of the same formulas (libfgs_oracle64.so = fgs_oracle.c built with -DORC_F64) ---------------------
_LIB64 = None


class _Settings64(C.Structure):
    _fields_ = [
        ('w2c', C.c_double * 12), ('cam_pos', C.c_double * 3), ('bg', C.c_double * 3),
        ('active_sh_bases', C.c_int), ('total_sh_rest', C.c_int), ('width', C.c_int), ('height', C.c_int),
        ('fx', C.c_double), ('fy', C.c_double), ('cx', C.c_double), ('cy', C.c_double),
        ('near_plane', C.c_double), ('far_plane', C.c_double), ('proper_aa', C.c_int),
    ]


def lib64() -> C.CDLL:
    global _LIB64
    if _LIB64 is None:
        build()
        _LIB64 = C.CDLL(str(_HERE / 'libfgs_oracle64.so'))
    return _LIB64


def forward_backward_f64(fwd: dict, settings: Settings, grad_image) -> dict:
    """The image and the six gradients of `fwd`'s scene evaluated in DOUBLE precision: the same formulas in the same order
    (fgs_oracle.c built with -DORC_F64), on the fp32 inputs and the fp32 constants, over the discrete structure of the fp32 run `fwd`
    (visible set, bounds, instance lists, ranges, buckets). The per-pair decisions (alpha >= 1/255, T < 1e-4) are re-taken in double;
    the few that land on the other side are the pairs oracle.threshold_risk names. Returns {'image', 'final_T', 'means', 'scales',
    'rotations', 'opacities', 'sh0', 'sh_rest'} as float64 arrays -- the tests' estimate of the true values."""
    L = lib64()
    f64 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float64)
    means, scales, rotations, opacities, sh0, sh_rest = (f64(a) for a in fwd['_inputs'])
    N, B, T, bs = fwd['N'], fwd['B'], fwd['T'], fwd['bucket_size']
    total_rest = sh_rest.shape[1] if sh_rest.ndim == 3 else 0
    S32 = fwd['_S']
    S = _Settings64()
    for name, _ in _Settings64._fields_:
        v = getattr(S32, name)
        if hasattr(v, '__len__'):
            getattr(S, name)[:] = [float(x) for x in v]
        else:
            setattr(S, name, v)
    S.total_sh_rest = int(total_rest)
    W, H = settings.width, settings.height
    P = W * H
    mean2d, conic, color = np.zeros((N, 2)), np.zeros((N, 4)), np.zeros((N, 3))
    L.orc_preprocess_follow(N, _p(means), _p(scales), _p(rotations), _p(opacities), _p(sh0), _p(sh_rest), C.byref(S),
                            _p(fwd['n_touched']), _p(mean2d), _p(conic), _p(color))
    inst_prims = np.ascontiguousarray(fwd['inst_prims']) if fwd['I'] > 0 else np.zeros(1, np.uint32)
    image, final_T = np.zeros((3, H, W)), np.ones(P)
    n_processed, max_n_processed = np.zeros(P, np.uint32), np.zeros(T, np.uint32)
    bti = np.zeros(max(B, 1), np.uint32)
    ckpt = np.zeros((max(B, 1), BLOCK_BLEND, 4))
    L.orc_blend_forward(0, 1, 0, bs, _p(fwd['ranges']), _p(fwd['bucket_offsets']), _p(inst_prims), _p(fwd['screen_bounds']), _p(mean2d),
                        _p(conic), _p(color), C.byref(S), _p(image), _p(final_T), _p(n_processed), _p(max_n_processed), _p(bti), _p(ckpt))
    gi = f64(grad_image).reshape(3, H, W)
    g = {'means': np.zeros((N, 3)), 'scales': np.zeros((N, 3)), 'rotations': np.zeros((N, 4)), 'opacities': np.zeros((N, 1)),
         'sh0': np.zeros((N, 1, 3)), 'sh_rest': np.zeros(sh_rest.shape)}
    grad_mean2d, grad_conic = np.zeros((N, 2)), np.zeros((3, N))
    L.orc_blend_backward(N, B, bs, _p(fwd['ranges']), _p(fwd['bucket_offsets']), _p(inst_prims), _p(mean2d), _p(conic), _p(color),
                         C.byref(S), _p(gi), _p(image), _p(final_T), _p(max_n_processed), _p(n_processed), _p(bti), _p(ckpt),
                         _p(grad_mean2d), _p(grad_conic), _p(g['opacities']), _p(g['sh0']))
    L.orc_preprocess_backward(N, _p(means), _p(scales), _p(rotations), _p(opacities), _p(sh_rest), C.byref(S), _p(fwd['n_touched']),
                              _p(grad_mean2d), _p(grad_conic), _p(g['means']), _p(g['scales']), _p(g['rotations']), _p(g['opacities']),
                              _p(g['sh0']), _p(g['sh_rest']), None)
    g.update(image=image, final_T=final_T, n_processed=n_processed)
    return g


def records_f64(fwd: dict, settings: Settings) -> dict:
    """Test support: K1's per-Gaussian records (mean2d, conic + opacity, colour) evaluated in DOUBLE on the fp32 inputs, for the Gaussians the fp32 run found
    visible (orc_preprocess_follow, as forward_backward_f64 uses it) -- the fp64 reference of a conditioning-aware comparison of two fp32 K1s."""
    L = lib64()
    f64 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float64)
    means, scales, rotations, opacities, sh0, sh_rest = (f64(a) for a in fwd['_inputs'])
    N = fwd['N']
    S32 = fwd['_S']
    S = _Settings64()
    for name, _ in _Settings64._fields_:
        v = getattr(S32, name)
        if hasattr(v, '__len__'):
            getattr(S, name)[:] = [float(x) for x in v]
        else:
            setattr(S, name, v)
    S.total_sh_rest = int(sh_rest.shape[1] if sh_rest.ndim == 3 else 0)
    mean2d, conic, color = np.zeros((N, 2)), np.zeros((N, 4)), np.zeros((N, 3))
    L.orc_preprocess_follow(N, _p(means), _p(scales), _p(rotations), _p(opacities), _p(sh0), _p(sh_rest), C.byref(S),
                            _p(fwd['n_touched']), _p(mean2d), _p(conic), _p(color))
    return {'mean2d': mean2d, 'conic_opacity': conic, 'color': color}


def blend_sums_f64(fwd: dict, settings: Settings, grad_image) -> dict:
    """Test support: the blend forward and K11's nine per-Gaussian sums in DOUBLE on the records held by `fwd` (reblend's output, say) over its discrete
    structure -- the fp64 reference of helpers.check_blend_on_device_records' three-way bars. The per-pair decisions are re-taken in double, as in
    forward_backward_f64. Returns {'image', 'final_T', 'sums' [N, 9] (mean2d.xy, conic.abc, opacity, colour.rgb)}."""
    L = lib64()
    f64 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float64)
    N, B, T, bs = fwd['N'], fwd['B'], fwd['T'], fwd['bucket_size']
    S32 = fwd['_S']
    S = _Settings64()
    for name, _ in _Settings64._fields_:
        v = getattr(S32, name)
        if hasattr(v, '__len__'):
            getattr(S, name)[:] = [float(x) for x in v]
        else:
            setattr(S, name, v)
    W, H = settings.width, settings.height
    mean2d, conic, color = f64(fwd['mean2d']), f64(fwd['conic_opacity']), f64(fwd['color'])
    inst_prims = np.ascontiguousarray(fwd['inst_prims']) if fwd['I'] > 0 else np.zeros(1, np.uint32)
    image, final_T = np.zeros((3, H, W)), np.ones(W * H)
    n_processed, max_n_processed = np.zeros(W * H, np.uint32), np.zeros(T, np.uint32)
    bti = np.zeros(max(B, 1), np.uint32)
    ckpt = np.zeros((max(B, 1), BLOCK_BLEND, 4))
    L.orc_blend_forward(0, 1, 0, bs, _p(fwd['ranges']), _p(fwd['bucket_offsets']), _p(inst_prims), _p(fwd['screen_bounds']), _p(mean2d),
                        _p(conic), _p(color), C.byref(S), _p(image), _p(final_T), _p(n_processed), _p(max_n_processed), _p(bti), _p(ckpt))
    gi = f64(grad_image).reshape(3, H, W)
    grad_mean2d, grad_conic, g_op, g_col = np.zeros((N, 2)), np.zeros((3, N)), np.zeros((N, 1)), np.zeros((N, 1, 3))
    L.orc_blend_backward(N, B, bs, _p(fwd['ranges']), _p(fwd['bucket_offsets']), _p(inst_prims), _p(mean2d), _p(conic), _p(color),
                         C.byref(S), _p(gi), _p(image), _p(final_T), _p(max_n_processed), _p(n_processed), _p(bti), _p(ckpt),
                         _p(grad_mean2d), _p(grad_conic), _p(g_op), _p(g_col))
    return {'image': image, 'final_T': final_T, 'sums': np.concatenate([grad_mean2d, grad_conic.T, g_op.reshape(N, 1), g_col.reshape(N, 3)], axis=1)}


def threshold_risk(fwd: dict, settings: Settings, eps: float = 1e-5, eps_T: float = 1e-4) -> dict:
    """Test support (fgs_oracle.c: orc_threshold_risk): boolean masks of the pixels / Gaussians whose blend decisions lie within
    `eps` (relative) of the alpha >= 1/255 threshold (or within eps_T of the T < 1e-4 termination), plus the Gaussians that
    blend into such a pixel. Parity tests exclude and count exactly these entries."""
    W, H, N = settings.width, settings.height, fwd['N']
    risk_pixel = np.zeros(W * H, np.uint8)
    risk_prim = np.zeros(max(N, 1), np.uint8)
    near_prim = np.zeros(max(N, 1), np.uint8)
    inst_prims = np.ascontiguousarray(fwd['inst_prims']) if fwd['I'] > 0 else np.zeros(1, np.uint32)
    lib().orc_threshold_risk(_p(fwd['ranges']), _p(inst_prims), _p(fwd['screen_bounds']), _p(fwd['mean2d']), _p(fwd['conic_opacity']),
                             C.byref(fwd['_S']), _p(fwd['n_processed']), C.c_float(eps), C.c_float(eps_T),
                             _p(risk_pixel), _p(risk_prim), _p(near_prim))
    return {'pixel': risk_pixel.reshape(H, W).astype(bool), 'prim': risk_prim[:N].astype(bool), 'near': near_prim[:N].astype(bool)}


def adam_step(grad: np.ndarray, param: np.ndarray, exp_avg: np.ndarray, exp_avg_sq: np.ndarray, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15) -> None:
    """In-place Adam update (adam.cu:10-71)."""
    for a in (grad, param, exp_avg, exp_avg_sq):
        assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    lib().orc_adam_step(_p(grad), _p(param), _p(exp_avg), _p(exp_avg_sq), C.c_longlong(param.size), int(step),
                        C.c_double(lr), C.c_double(beta1), C.c_double(beta2), C.c_double(eps))


def l1_dssim(image: np.ndarray, target: np.ndarray, lambda_l1: float = 0.8, lambda_dssim: float = 0.2, with_grad: bool = True):
    """Loss of Trainer.py:190-193 / Loss.py:15-16 (restated SSIM, see fgs_oracle.c). Returns (loss, l1, ssim, grad or None)."""
    L = lib()
    L.orc_l1_dssim.restype = C.c_float
    image, target = _f32(image), _f32(target)
    _, H, W = image.shape
    grad = np.zeros_like(image) if with_grad else None
    l1, ssim = C.c_float(0), C.c_float(0)
    loss = L.orc_l1_dssim(_p(image), _p(target), H, W, C.c_float(lambda_l1), C.c_float(lambda_dssim),
                          _p(grad) if with_grad else None, C.byref(l1), C.byref(ssim))
    return float(loss), float(l1.value), float(ssim.value), grad


def pruning_scores(scores: np.ndarray, means, scales, rotations, opacities, sh0, sh_rest, settings: Settings) -> dict:
    """update_pruning_scores (rasterization.py:159-178 -> pruning_scores.cu): accumulates into `scores` [N] in place."""
    assert scores.dtype == np.float32 and scores.flags['C_CONTIGUOUS']
    f = forward(means, scales, rotations, opacities, sh0, sh_rest, settings, inference=True)
    S = settings.to_c(np.asarray(sh_rest).reshape(f['N'], -1, 3).shape[1] if f['N'] else 0)
    inst = np.ascontiguousarray(f['inst_prims']) if f['I'] > 0 else np.zeros(1, np.uint32)
    lib().orc_pruning_scores(_p(f['ranges']), _p(inst), _p(f['screen_bounds']), _p(f['mean2d']), _p(f['conic_opacity']),
                             _p(f['color']), C.byref(S), f['N'], _p(scores))
    return f


# ---- the three small exported operators, restated in numpy fp32 ------------------------------------
def update_3d_filter(positions, w2c, filter_3d, visibility_mask, width, height, focal_x, focal_y, center_x, center_y,
                     near_plane, clipping_tolerance, distance2filter) -> None:
    """filter3d/src/filter3d.cu:9-83; filter_3d / visibility_mask are updated in place."""
    f = np.float32
    p, w = _f32(positions), _f32(w2c).reshape(-1)[:12]
    bounds = f(clipping_tolerance) + f(0.5)
    wf, hf = f(width), f(height)
    mx, my = bounds * wf, bounds * hf
    ox, oy = f(center_x) - f(0.5) * wf, f(center_y) - f(0.5) * hf
    left, right = (-mx - ox) / f(focal_x), (mx - ox) / f(focal_x)
    top, bottom = (-my - oy) / f(focal_y), (my - oy) / f(focal_y)
    z = (w[8] * p[:, 0] + w[9] * p[:, 1] + w[10] * p[:, 2]) + w[11]
    xc = (w[0] * p[:, 0] + w[1] * p[:, 1] + w[2] * p[:, 2]) + w[3]
    yc = (w[4] * p[:, 0] + w[5] * p[:, 1] + w[6] * p[:, 2]) + w[7]
    ok = (z >= f(near_plane)) & ~((xc < left * z) | (xc > right * z)) & ~((yc < top * z) | (yc > bottom * z))
    new = f(distance2filter) * z
    upd = ok & ~(filter_3d < new)
    filter_3d[upd] = new[upd]
    visibility_mask[upd] = True


def relocation_adjustment(old_opacities, old_scales, n_samples_per_primitive):
    """densification/include/kernels_mcmc.cuh:10-59 (Eq. 9 of 3DGS-MCMC), max 50 samples."""
    f = np.float32
    op, sc = _f32(old_opacities).reshape(-1), _f32(old_scales).reshape(-1, 3)
    ns = np.clip(np.asarray(n_samples_per_primitive).astype(np.int64), 1, 50)
    table = np.zeros((50, 50), np.float32)
    for n in range(50):
        binom, sign = 1.0, 1.0
        for k in range(n + 1):
            table[n, k] = f(binom * sign / np.sqrt(float(k + 1)))
            binom *= (n - k) / (k + 1)
            sign = -sign
    new_op = (f(1.0) - np.power(f(1.0) - op, f(1.0) / ns.astype(np.float32))).astype(np.float32)
    new_sc = np.zeros_like(sc)
    for i in range(op.shape[0]):
        den = f(0.0)
        for n in range(int(ns[i])):
            power = new_op[i]
            for k in range(n + 1):
                den = f(den + table[n, k] * power)
                power = f(power * new_op[i])
        new_sc[i] = (op[i] / den) * sc[i]
    return new_op.reshape(-1, 1), new_sc


def add_noise(raw_scales, raw_rotations, raw_opacities, random_samples, means, current_lr) -> None:
    """densification/include/kernels_mcmc.cuh:69-127; `means` [N,3] float32 is updated in place."""
    f = np.float32
    s, q, o, r = _f32(raw_scales), _f32(raw_rotations), _f32(raw_opacities).reshape(-1), _f32(random_samples)
    var = np.exp(f(2.0) * s)
    rr, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    nsq = rr * rr + x * x + y * y + z * z
    ok = nsq >= f(1e-8)
    inv = f(1.0) / np.where(ok, nsq, f(1.0))
    R = np.stack([f(1) - f(2) * (y * y + z * z) * inv, f(2) * (x * y - rr * z) * inv, f(2) * (x * z + rr * y) * inv,
                  f(2) * (x * y + rr * z) * inv, f(1) - f(2) * (x * x + z * z) * inv, f(2) * (y * z - rr * x) * inv,
                  f(2) * (x * z - rr * y) * inv, f(2) * (y * z + rr * x) * inv, f(1) - f(2) * (x * x + y * y) * inv], 1).reshape(-1, 3, 3)
    cov = (R * var[:, None, :]) @ R.transpose(0, 2, 1)
    t = np.einsum('nij,nj->ni', cov, r).astype(np.float32)
    opacity = f(1.0) / (f(1.0) + np.exp(-o))
    factor = f(current_lr) * (f(1.0) / (f(1.0) + np.exp(f(100.0) * opacity - f(0.5))))
    means[ok] += (factor[:, None] * t)[ok]


# ---- maintenance of the Gaussian set, restated in numpy fp32 -------------------------------------------------
def adaptive_density_control(params: dict, exp_avgs: dict | None, exp_avg_sqs: dict | None, densification_info: np.ndarray, noise: np.ndarray,
                             grad_threshold: float, min_opacity: float, prune_large_gaussians: bool, percent_dense: float, extent: float):
    """Model.py:312-366 line by line (with extend_param_groups / prune_param_groups of the un-vendored NeRFICG Optim.adam_utils: new
    entries get zero moments, survivors keep theirs). params / moments: dicts of float32 arrays keyed means, sh_coefficients_0,
    sh_coefficients_rest, opacities, scales, rotations; noise: the [2 * n_split, 3] samples of Model.py:331's randn_like.
    Returns (new_params, new_exp_avgs, new_exp_avg_sqs, counts) with counts = (survivors, surviving clones, surviving children per copy, split)."""
    f = np.float32
    keys = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
    P = {k: _f32(params[k]) for k in keys}
    info = _f32(densification_info)
    scales, rotations = P['scales'], P['rotations']
    densification_mask = info[1] >= f(grad_threshold) * np.maximum(info[0], f(1.0))                                         # :314
    is_small = scales.max(axis=1) <= f(np.log(percent_dense * extent))                                                        # :315
    duplicate_mask = densification_mask & is_small                                                                            # :318
    split_mask = densification_mask & ~is_small                                                                               # :328
    n_split = int(split_mask.sum())
    split_scales = np.tile(np.exp(scales[split_mask]), (2, 1))                                                                # :330  expand(2,...).flatten
    split_rotations = np.tile(rotations[split_mask], (2, 1))                                                                  # :331
    q = split_rotations / np.sqrt((split_rotations * split_rotations).sum(axis=1, keepdims=True))                             # quaternion_to_rotation_matrix
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([f(1) - f(2) * (y * y + z * z), f(2) * (x * y - r * z), f(2) * (x * z + r * y),
                  f(2) * (x * y + r * z), f(1) - f(2) * (x * x + z * z), f(2) * (y * z - r * x),
                  f(2) * (x * z - r * y), f(2) * (y * z + r * x), f(1) - f(2) * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    t = split_scales * _f32(noise).reshape(2 * n_split, 3)
    offsets = (R[:, :, 0] * t[:, None, 0] + R[:, :, 1] * t[:, None, 1]) + R[:, :, 2] * t[:, None, 2]                         # :332 (R @ t)
    extra = {
        'means': np.concatenate([P['means'][duplicate_mask], np.tile(P['means'][split_mask], (2, 1)) + offsets]),          # :333
        'sh_coefficients_0': np.concatenate([P['sh_coefficients_0'][duplicate_mask], np.tile(P['sh_coefficients_0'][split_mask], (2, 1, 1))]),
        'sh_coefficients_rest': np.concatenate([P['sh_coefficients_rest'][duplicate_mask], np.tile(P['sh_coefficients_rest'][split_mask], (2, 1, 1))]),
        'opacities': np.concatenate([P['opacities'][duplicate_mask], np.tile(P['opacities'][split_mask], (2, 1))]),
        'scales': np.concatenate([scales[duplicate_mask], np.log(split_scales * f(0.625))]),                                 # :334
        'rotations': np.concatenate([rotations[duplicate_mask], split_rotations]),
    }
    n_new = extra['means'].shape[0]
    full = {k: np.concatenate([P[k], extra[k]]) for k in keys}                                                                # :340-347 extend_param_groups
    prune_mask = np.concatenate([split_mask, np.zeros(n_new, bool)])                                                          # :359
    prune_mask |= full['opacities'].reshape(-1) < f(np.log(min_opacity / (1 - min_opacity)))                                  # :360
    prune_mask |= (full['rotations'] * full['rotations']).sum(axis=1) < f(1e-8)                                               # :361
    if prune_large_gaussians:
        prune_mask |= full['scales'].max(axis=1) > f(np.log(0.1 * extent))                                                    # :362-363
    keep = ~prune_mask                                                                                                        # :364 prune()
    new_p = {k: np.ascontiguousarray(full[k][keep]) for k in keys}
    new_m = new_v = None
    if exp_avgs is not None:
        new_m = {k: np.ascontiguousarray(np.concatenate([_f32(exp_avgs[k]), np.zeros_like(extra[k])])[keep]) for k in keys}
        new_v = {k: np.ascontiguousarray(np.concatenate([_f32(exp_avg_sqs[k]), np.zeros_like(extra[k])])[keep]) for k in keys}
    n_old, n_dup = P['means'].shape[0], int(duplicate_mask.sum())
    counts = (int(keep[:n_old].sum()), int(keep[n_old:n_old + n_dup].sum()), int(keep[n_old + n_dup:n_old + n_dup + n_split].sum()), n_split)
    return new_p, new_m, new_v, counts


def morton_order(means: np.ndarray) -> np.ndarray:
    """Model.py:459-463 with the 30-bit Z-curve that stands in for NeRFICG's CudaUtils.MortonEncoding (harness/scenes.py:67-80):
    10 bits per axis over the bounding box, x most significant, stable argsort."""
    f = np.float32
    m = _f32(means)
    lo, hi = m.min(axis=0), m.max(axis=0)
    q = ((m - lo) / np.maximum(hi - lo, f(1e-12)) * f(1023.0)).astype(np.int64).clip(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    return np.argsort(code, kind='stable')
