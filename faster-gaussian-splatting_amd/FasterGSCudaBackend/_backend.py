"""torch <-> C-ABI glue: turns tensors into raw pointers, owns the resize callback, checks status codes.

One `Backend` wraps one loaded library handle. The product uses `default_backend()` (libfgs_hip.so); the test-suite also
builds one around the CPU simulation library to exercise this exact code path without a GPU.
Mirrors what the reference does in C++ in rasterization_api.cu:13-247 and utils/torch_utils.h:6-12.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Sequence

import torch

from . import _lib


class RasterizerSettings(NamedTuple):
    """Same 13 fields, order and meaning as the reference (torch_bindings/rasterization.py:8-38)."""
    w2c: torch.Tensor            # affine transformation from model/world space to view space (row-major, >= 3 rows)
    cam_position: torch.Tensor   # camera position in world space
    bg_color: torch.Tensor       # background colour (RGB)
    active_sh_bases: int         # number of SH bases used for colour
    width: int                   # image width in pixels
    height: int                  # image height in pixels
    focal_x: float               # focal length in pixels
    focal_y: float
    center_x: float              # principal point in pixels, +x right
    center_y: float              # +y down
    near_plane: float
    far_plane: float
    proper_antialiasing: bool

    def as_tuple(self) -> tuple:
        return tuple(self)


class ForwardResult(NamedTuple):
    image: torch.Tensor
    buffers: tuple          # (primitive, tile, instance, bucket) uint8 tensors, opaque
    state: tuple            # (n_visible, n_instances, n_buckets, selector), opaque


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _stream_of(device: torch.device) -> int:
    """The caller's current stream on the tensors' device. The library launches on the CURRENT device (it never switches
    devices), so a mismatch is reported here instead of surfacing as an opaque HIP error later."""
    if device.type != 'cuda':
        return 0
    if device.index is not None and device.index != torch.cuda.current_device():
        raise RuntimeError(f'tensors live on {device} but the current device is cuda:{torch.cuda.current_device()}; '
                           f'call under torch.cuda.device({device.index})')
    return torch.cuda.current_stream(device).cuda_stream


class Backend:
    def __init__(self, lib: C.CDLL):
        self.lib = lib

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _check(self, status: int, what: str) -> None:
        if status != 0:
            raise RuntimeError(f'{what} failed (status {status}): {self.lib.fgs_last_error().decode()}')

    @staticmethod
    def _settings(s: RasterizerSettings, total_sh_rest: int, device: torch.device, keep: list) -> _lib.Settings:
        w2c = s.w2c.to(device=device, dtype=torch.float32).contiguous()
        cam = s.cam_position.to(device=device, dtype=torch.float32).contiguous()
        bg = s.bg_color.to(device=device, dtype=torch.float32).contiguous()
        if w2c.numel() < 12 or cam.numel() < 3 or bg.numel() < 3:
            raise ValueError('w2c needs >= 3x4, cam_position and bg_color 3 entries')
        keep += [w2c, cam, bg]
        return _lib.Settings(w2c.data_ptr(), cam.data_ptr(), bg.data_ptr(), int(s.active_sh_bases), int(total_sh_rest),
                             int(s.width), int(s.height), float(s.focal_x), float(s.focal_y), float(s.center_x),
                             float(s.center_y), float(s.near_plane), float(s.far_plane), int(bool(s.proper_antialiasing)))

    @staticmethod
    def _check_params(tensors: Sequence[torch.Tensor], names: Sequence[str]) -> torch.device:
        device = tensors[0].device
        for t, n in zip(tensors, names):   # utils/torch_utils.h:14-19 (CHECK_INPUT)
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
                raise RuntimeError(f"Input tensor '{n}' must be a contiguous float32 tensor on {device}.")
        return device

    @staticmethod
    def _make_resizer(device: torch.device, n_buffers: int):
        buffers = [torch.empty(0, dtype=torch.uint8, device=device) for _ in range(n_buffers)]

        def resize(_user, which, nbytes):          # utils/torch_utils.h:6-12
            try:
                buffers[which].resize_(int(nbytes))
                return buffers[which].data_ptr() if nbytes else 0
            except Exception:                       # never let an exception cross the C boundary
                return 0
        return buffers, _lib.RESIZE_FN(resize)

    # -- entry points -----------------------------------------------------------------------------------------------
    def forward(self, means, scales, rotations, opacities, sh0, sh_rest, settings: RasterizerSettings,
                instance_capacity: int | None = None) -> ForwardResult:
        """instance_capacity: None = fgs_forward (one host read of the counts); an int = fgs_forward_async, no host wait -- state then
        holds bounds (N, capacity, ...) and `forward_counts` tells later whether the capacity was enough."""
        device = self._check_params((means, scales, rotations, opacities, sh0, sh_rest),
                                    ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest'))
        keep: list = []
        S = self._settings(settings, sh_rest.shape[1] if sh_rest.dim() == 3 else 0, device, keep)
        image = torch.empty((3, settings.height, settings.width), dtype=torch.float32, device=device)
        buffers, cb = self._make_resizer(device, 4)
        st = _lib.ForwardState()
        if instance_capacity is None:
            self._check(self.lib.fgs_forward(_ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(sh0), _ptr(sh_rest),
                                             means.shape[0], C.byref(S), image.data_ptr(), cb, None, C.byref(st), _stream_of(device)),
                        'fgs_forward')
        else:
            self._check(self.lib.fgs_forward_async(_ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(sh0), _ptr(sh_rest),
                                                   means.shape[0], C.byref(S), image.data_ptr(), int(instance_capacity), cb, None, C.byref(st),
                                                   _stream_of(device)), 'fgs_forward_async')
        return ForwardResult(image, tuple(buffers), (st.n_visible, st.n_instances, st.n_buckets, st.selector))

    def forward_counts(self, result: ForwardResult, n_primitives: int):
        """Enqueues the read-back of (n_visible, n_instances, capacity_exceeded) of a forward pass; returns (pinned int32[3] tensor, event).
        The values are valid after event.synchronize() -- typically free by the time backward runs."""
        host = torch.empty(3, dtype=torch.int32, pin_memory=result.image.is_cuda)
        self._check(self.lib.fgs_forward_counts(_ptr(result.buffers[0]), int(n_primitives), host.data_ptr(), _stream_of(result.image.device)),
                    'fgs_forward_counts')
        event = None
        if result.image.is_cuda:
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(result.image.device))
            # The copy is a raw hipMemcpyAsync that torch's caching host allocator knows nothing about: if the caller drops `host` before the
            # copy has run (forward under no_grad, an exception, a freed graph), the pinned block would go back to the cache and the late copy
            # would land in whoever got it next. Keep every buffer referenced here until its event has completed.
            pending = self.__dict__.setdefault('_pending_counts', [])
            pending[:] = [(h, e) for h, e in pending if not e.query()]
            pending.append((host, event))
        return host, event

    def inference(self, means, scales, rotations, opacities, sh0, sh_rest, settings: RasterizerSettings, to_chw: bool,
                  clamp_output: bool, return_state: bool = False):
        device = self._check_params((means, scales, rotations, opacities, sh0, sh_rest),
                                    ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest'))
        keep: list = []
        S = self._settings(settings, sh_rest.shape[1] if sh_rest.dim() == 3 else 0, device, keep)
        shape = (3, settings.height, settings.width) if to_chw else (settings.height, settings.width, 3)
        image = torch.empty(shape, dtype=torch.float32, device=device)
        buffers, cb = self._make_resizer(device, 4)
        st = _lib.ForwardState()
        self._check(self.lib.fgs_inference(_ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(sh0), _ptr(sh_rest),
                                           means.shape[0], C.byref(S), image.data_ptr(), int(to_chw), int(clamp_output), cb, None,
                                           C.byref(st), _stream_of(device)), 'fgs_inference')
        if return_state:
            return ForwardResult(image, tuple(buffers), (st.n_visible, st.n_instances, st.n_buckets, st.selector))
        return image

    def pruning_scores(self, scores, means, scales, rotations, opacities, sh0, sh_rest, settings: RasterizerSettings) -> None:
        """Accumulates the Speedy-Splat importance scores of one view into `scores` [N] (rasterization.py:159-178)."""
        device = self._check_params((scores, means, scales, rotations, opacities, sh0, sh_rest),
                                    ('scores', 'means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest'))
        if scores.numel() != means.shape[0]:
            raise RuntimeError('scores must have one entry per Gaussian')
        keep: list = []
        S = self._settings(settings, sh_rest.shape[1] if sh_rest.dim() == 3 else 0, device, keep)
        buffers, cb = self._make_resizer(device, 4)
        st = _lib.ForwardState()
        self._check(self.lib.fgs_pruning_scores(_ptr(scores), _ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(sh0),
                                                _ptr(sh_rest), means.shape[0], C.byref(S), cb, None, C.byref(st), _stream_of(device)),
                    'fgs_pruning_scores')

    def _scratch(self, n: int, settings: RasterizerSettings, device: torch.device) -> torch.Tensor:
        nbytes = int(self.lib.fgs_backward_scratch_bytes(n, int(settings.width), int(settings.height)))
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)

    def backward(self, densification_info, grad_image, image, means, scales, rotations, opacities, sh_rest, buffers, settings,
                 state, out: tuple | None = None, live_blocks: Optional[torch.Tensor] = None) -> tuple:
        """`out`: optional six preallocated gradient tensors (e.g. views into one contiguous arena for a single RCCL call).
        `live_blocks`: optional uint8 [ceil(N / 64)] on the device, filled with 1 / 0 per block of 64 Gaussians: 0 = every gradient of the block
        is zero (still written) -- what adam_step_multi(live_blocks=...) needs to skip reading those zeros."""
        device = self._check_params((means, scales, rotations, opacities, sh_rest), ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_rest'))
        keep: list = []
        n = means.shape[0]
        total_rest = sh_rest.shape[1] if sh_rest.dim() == 3 else 0
        S = self._settings(settings, total_rest, device, keep)
        grad_image = grad_image.to(dtype=torch.float32).contiguous()
        shapes = ((n, 3), (n, 3), (n, 4), (n, 1), (n, 1, 3), (n, total_rest, 3))
        if out is None:
            grads = tuple(torch.empty(sh, dtype=torch.float32, device=device) for sh in shapes)
        else:
            grads = tuple(out)
            for g, sh in zip(grads, shapes):
                if tuple(g.shape) != sh or g.dtype != torch.float32 or not g.is_contiguous() or g.device != device:
                    raise RuntimeError(f'preallocated gradient has shape {tuple(g.shape)}, expected contiguous float32 {sh}')
        dens = densification_info if densification_info is not None and densification_info.numel() > 0 else None   # api:136
        if dens is not None and (dens.dtype != torch.float32 or not dens.is_contiguous() or dens.device != device or dens.numel() != 2 * n):
            raise RuntimeError('densification_info must be a contiguous float32 [2, N] tensor on the parameters\' device')
        scratch = self._scratch(n, settings, device)
        st = _lib.ForwardState(*state)
        if live_blocks is not None and (live_blocks.dtype != torch.uint8 or live_blocks.device != device or not live_blocks.is_contiguous()
                                        or live_blocks.numel() != (n + 63) // 64):
            raise RuntimeError('live_blocks must be a contiguous uint8 tensor of ceil(N / 64) elements on the parameters\' device')
        self._check(self.lib.fgs_backward_live(_ptr(grad_image), _ptr(image), _ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities),
                                               _ptr(sh_rest), _ptr(buffers[0]), _ptr(buffers[1]), _ptr(buffers[2]), _ptr(buffers[3]),
                                               _ptr(grads[0]), _ptr(grads[1]), _ptr(grads[2]), _ptr(grads[3]), _ptr(grads[4]), _ptr(grads[5]),
                                               _ptr(dens), scratch.data_ptr(), n, C.byref(S), C.byref(st), _ptr(live_blocks), _stream_of(device)),
                    'fgs_backward')
        return grads

    def backward_adam_fused(self, densification_info, grad_image, image, params: Sequence[torch.Tensor], exp_avgs, exp_avg_sqs,
                            buffers, settings, state, step: int, lrs: Sequence[float], betas=(0.9, 0.999), eps: float = 1e-15) -> None:
        """params / moments / lrs in optimizer-group order: means, sh0, sh_rest, opacities, scales, rotations (Model.py:238-245)."""
        device = self._check_params(tuple(params) + tuple(exp_avgs) + tuple(exp_avg_sqs), ['param/moment'] * 18)
        keep: list = []
        n = params[0].shape[0]
        total_rest = params[2].shape[1] if params[2].dim() == 3 else 0
        S = self._settings(settings, total_rest, device, keep)
        grad_image = grad_image.to(dtype=torch.float32).contiguous()
        dens = densification_info if densification_info is not None and densification_info.numel() > 0 else None
        scratch = self._scratch(n, settings, device)
        st = _lib.ForwardState(*state)
        arr = lambda ts: (C.c_void_p * 6)(*[_ptr(t) for t in ts])
        lr_arr = (C.c_double * 6)(*[float(x) for x in lrs])
        self._check(self.lib.fgs_backward_adam_fused(_ptr(grad_image), _ptr(image), arr(params), arr(exp_avgs), arr(exp_avg_sqs),
                                                     _ptr(buffers[0]), _ptr(buffers[1]), _ptr(buffers[2]), _ptr(buffers[3]), _ptr(dens),
                                                     scratch.data_ptr(), n, C.byref(S), C.byref(st), int(step), lr_arr,
                                                     float(betas[0]), float(betas[1]), float(eps), _stream_of(device)),
                    'fgs_backward_adam_fused')

    def adam_step(self, grad, param, exp_avg, exp_avg_sq, step: int, lr: float, beta1: float, beta2: float, eps: float) -> None:
        device = self._check_params((grad, param, exp_avg, exp_avg_sq), ('param_grad', 'param', 'exp_avg', 'exp_avg_sq'))
        self._check(self.lib.fgs_adam_step(_ptr(grad), _ptr(param), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), int(step),
                                           float(lr), float(beta1), float(beta2), float(eps), _stream_of(device)), 'fgs_adam_step')

    def adam_step_multi(self, grads, params, exp_avgs, exp_avg_sqs, steps, lrs, beta1: float, beta2: float, eps: float,
                        live_blocks: Optional[torch.Tensor] = None) -> None:
        """`live_blocks` (as filled by backward(live_blocks=...) for EXACTLY these gradient tensors, all of them [N, ...]): a promise that the
        gradient rows of blocks flagged 0 are zero; they are not read. The result is bit-identical either way."""
        k = len(params)
        if k == 0:
            return
        device = self._check_params(tuple(grads) + tuple(params) + tuple(exp_avgs) + tuple(exp_avg_sqs), ['adam tensor'] * (4 * k))
        arr = lambda ts: (C.c_void_p * k)(*[_ptr(t) for t in ts])
        rows = None
        if live_blocks is not None:
            n = params[0].shape[0]
            if any(p.dim() < 1 or p.shape[0] != n for p in params) or n == 0:
                raise RuntimeError('live_blocks needs parameter tensors that all have one row per Gaussian')
            if live_blocks.dtype != torch.uint8 or live_blocks.device != device or not live_blocks.is_contiguous() or live_blocks.numel() != (n + 63) // 64:
                raise RuntimeError('live_blocks must be a contiguous uint8 tensor of ceil(N / 64) elements on the parameters\' device')
            rows = (C.c_int32 * k)(*[p.numel() // n for p in params])
        self._check(self.lib.fgs_adam_step_multi_live(k, arr(grads), arr(params), arr(exp_avgs), arr(exp_avg_sqs),
                                                      (C.c_int64 * k)(*[p.numel() for p in params]), (C.c_int32 * k)(*[int(s) for s in steps]),
                                                      (C.c_double * k)(*[float(x) for x in lrs]), float(beta1), float(beta2), float(eps),
                                                      _ptr(live_blocks), rows, _stream_of(device)), 'fgs_adam_step_multi')

    def l1_dssim(self, image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2,
                 with_grad: bool = True):
        """Fused photometric loss (Loss.py:15-16): returns (loss 0-dim tensor, dloss/dimage or None, (l1, ssim) tensor)."""
        device = self._check_params((image, target), ('image', 'target'))
        if image.dim() != 3 or image.shape[0] != 3 or image.shape != target.shape:
            raise RuntimeError('l1_dssim expects two [3,H,W] tensors')
        _, h, w = image.shape
        sums = torch.empty(3, dtype=torch.float32, device=device)
        grad = torch.empty_like(image) if with_grad else None
        scratch = torch.empty(int(self.lib.fgs_l1_dssim_scratch_bytes(w, h)), dtype=torch.uint8, device=device)
        self._check(self.lib.fgs_l1_dssim_loss(image.data_ptr(), target.data_ptr(), w, h, float(lambda_l1), float(lambda_dssim),
                                               sums.data_ptr(), _ptr(grad), scratch.data_ptr(), _stream_of(device)), 'fgs_l1_dssim_loss')
        return sums[2], grad, sums[:2]      # loss and the (l1, ssim) means are formed on the device by the reduce kernel

    def l1_dssim_forward(self, image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2):
        """The loss value alone: returns (loss 0-dim tensor, (l1, ssim) tensor, scratch). `scratch` holds the derivative maps that
        l1_dssim_backward turns into dloss/dimage -- the shape of an autograd loss node (forward saves, backward launches one kernel)."""
        device = self._check_params((image, target), ('image', 'target'))
        if image.dim() != 3 or image.shape[0] != 3 or image.shape != target.shape:
            raise RuntimeError('l1_dssim expects two [3,H,W] tensors')
        _, h, w = image.shape
        sums = torch.empty(3, dtype=torch.float32, device=device)
        scratch = torch.empty(int(self.lib.fgs_l1_dssim_scratch_bytes(w, h)), dtype=torch.uint8, device=device)
        self._check(self.lib.fgs_l1_dssim_loss(image.data_ptr(), target.data_ptr(), w, h, float(lambda_l1), float(lambda_dssim),
                                               sums.data_ptr(), None, scratch.data_ptr(), _stream_of(device)), 'fgs_l1_dssim_loss')
        return sums[2], sums[:2], scratch

    def l1_dssim_backward(self, image: torch.Tensor, target: torch.Tensor, scratch: torch.Tensor, upstream: Optional[torch.Tensor] = None,
                          lambda_l1: float = 0.8, lambda_dssim: float = 0.2) -> torch.Tensor:
        """dloss/dimage * upstream from the maps of l1_dssim_forward (same image / target). `upstream`: a float32 scalar tensor on the
        device (dL/dloss), folded into the kernel -- no `grad * upstream` pass over the image -- or None for 1."""
        device = self._check_params((image, target), ('image', 'target'))
        _, h, w = image.shape
        if scratch.device != device or scratch.numel() < int(self.lib.fgs_l1_dssim_scratch_bytes(w, h)):
            raise RuntimeError('l1_dssim_backward: scratch is not the buffer l1_dssim_forward returned for this image size')
        if upstream is not None:
            if upstream.device != device or upstream.dtype != torch.float32 or upstream.numel() != 1:
                raise RuntimeError('l1_dssim_backward: upstream must be one float32 on the device of the image')
            upstream = upstream.reshape(1).contiguous()
        grad = torch.empty_like(image)
        self._check(self.lib.fgs_l1_dssim_backward(image.data_ptr(), target.data_ptr(), w, h, float(lambda_l1), float(lambda_dssim),
                                                   _ptr(upstream), grad.data_ptr(), scratch.data_ptr(), _stream_of(device)), 'fgs_l1_dssim_backward')
        return grad

    # -- Gaussian-sharded multi-GPU path (include/fgs_hip.h, "Gaussian-sharded multi-GPU path") --------------------------------
    def _settings_array(self, views: Sequence[RasterizerSettings], total_sh_rest: int, device: torch.device, keep: list):
        arr = (_lib.Settings * len(views))()
        for i, v in enumerate(views):
            arr[i] = self._settings(v, total_sh_rest, device, keep)
        return arr

    def shard_preprocess(self, means, scales, rotations, opacities, sh0, sh_rest, views: Sequence[RasterizerSettings],
                         records: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
        """K1 over this rank's shard for all views of the step. `records`: uint8 [len(views), N, 56] (view v: the first
        counts[v, 0] records are filled), `counts`: int32 [len(views), 2] device tensor receiving (n_visible, n_instances).
        Returns the primitive buffer to hand to `shard_backward`."""
        device = self._check_params((means, scales, rotations, opacities, sh0, sh_rest),
                                    ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest'))
        n, k = means.shape[0], len(views)
        if records.dtype != torch.uint8 or records.numel() < k * n * _lib.SPLAT_RECORD_BYTES or counts.dtype != torch.int32 \
                or counts.numel() < 2 * k or records.device != device or counts.device != device or not records.is_contiguous() \
                or not counts.is_contiguous():
            raise RuntimeError('records must be uint8 [views, N, 56] and counts int32 [views, 2] on the parameters\' device')
        keep: list = []
        S = self._settings_array(views, sh_rest.shape[1] if sh_rest.dim() == 3 else 0, device, keep)
        buffers, cb = self._make_resizer(device, 4)
        self._check(self.lib.fgs_shard_preprocess(_ptr(means), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(sh0), _ptr(sh_rest), n, k,
                                                  S, _ptr(records), counts.data_ptr(), cb, None, _stream_of(device)), 'fgs_shard_preprocess')
        return buffers[0]

    def forward_from_records(self, records: torch.Tensor, n_records: int, n_instances: int, settings: RasterizerSettings,
                             total_sh_rest: int, shard_counts: Sequence[int] | None = None) -> ForwardResult:
        """`shard_counts`: the records are the concatenation of that many records per shard; the renderer then places them interleaved (Morton
        neighbourhood of strided owners restored, include/fgs_hip.h). Pass the same counts to `backward_to_records`."""
        device = records.device
        if records.dtype != torch.uint8 or not records.is_contiguous() or records.numel() < n_records * _lib.SPLAT_RECORD_BYTES:
            raise RuntimeError('records must be a contiguous uint8 tensor of n_records * 56 bytes')
        keep: list = []
        S = self._settings(settings, total_sh_rest, device, keep)
        image = torch.empty((3, settings.height, settings.width), dtype=torch.float32, device=device)
        buffers, cb = self._make_resizer(device, 4)
        st = _lib.ForwardState()
        counts = (C.c_int32 * len(shard_counts))(*[int(c) for c in shard_counts]) if shard_counts else None
        self._check(self.lib.fgs_forward_from_shard_records(_ptr(records), int(n_records), int(n_instances), counts, len(shard_counts) if shard_counts else 0,
                                                            C.byref(S), image.data_ptr(), cb, None, C.byref(st), _stream_of(device)), 'fgs_forward_from_shard_records')
        return ForwardResult(image, tuple(buffers), (st.n_visible, st.n_instances, st.n_buckets, st.selector))

    def backward_to_records(self, grad_image, image, buffers, settings: RasterizerSettings, state, total_sh_rest: int,
                            out: torch.Tensor | None = None, shard_counts: Sequence[int] | None = None) -> torch.Tensor:
        """K11 of a view rendered by `forward_from_records` -> float32 [n_records, 9] accumulator records."""
        device = image.device
        n = int(state[0])
        keep: list = []
        S = self._settings(settings, total_sh_rest, device, keep)
        grad_image = grad_image.to(dtype=torch.float32).contiguous()
        acc = out if out is not None else torch.empty((n, 9), dtype=torch.float32, device=device)
        if acc.dtype != torch.float32 or acc.numel() < 9 * n or not acc.is_contiguous() or acc.device != device:
            raise RuntimeError('accumulator records must be contiguous float32 [n_records, 9]')
        scratch = self._scratch(n, settings, device)
        st = _lib.ForwardState(*state)
        counts = (C.c_int32 * len(shard_counts))(*[int(c) for c in shard_counts]) if shard_counts else None
        self._check(self.lib.fgs_backward_to_shard_records(_ptr(grad_image), _ptr(image), _ptr(buffers[0]), _ptr(buffers[1]), _ptr(buffers[2]),
                                                           _ptr(buffers[3]), scratch.data_ptr(), _ptr(acc), n, counts, len(shard_counts) if shard_counts else 0,
                                                           C.byref(S), C.byref(st), _stream_of(device)), 'fgs_backward_to_shard_records')
        return acc

    def shard_backward(self, acc_records: torch.Tensor, n_visible: Sequence[int], primitive_buffer: torch.Tensor, densification_info, means,
                       scales, rotations, opacities, sh_rest, views: Sequence[RasterizerSettings], out: tuple) -> tuple:
        """K12 on the shard, gradients summed over `views`; `acc_records`: float32 [sum(n_visible), 9] in view order;
        `out` = the six gradient tensors (means, scales, rotations, opacities, sh0, sh_rest), every element written."""
        device = self._check_params((means, scales, rotations, opacities, sh_rest) + tuple(out),
                                    ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_rest') + ('grad',) * 6)
        n, k = means.shape[0], len(views)
        total_rest = sh_rest.shape[1] if sh_rest.dim() == 3 else 0
        shapes = ((n, 3), (n, 3), (n, 4), (n, 1), (n, 1, 3), (n, total_rest, 3))
        for g, sh in zip(out, shapes):
            if tuple(g.shape) != sh:
                raise RuntimeError(f'preallocated gradient has shape {tuple(g.shape)}, expected {sh}')
        total = int(sum(n_visible))
        if len(n_visible) != k or (total > 0 and (acc_records.dtype != torch.float32 or acc_records.numel() < 9 * total
                                                  or not acc_records.is_contiguous() or acc_records.device != device)):
            raise RuntimeError('acc_records must be contiguous float32 [sum(n_visible), 9] and n_visible one entry per view')
        keep: list = []
        S = self._settings_array(views, total_rest, device, keep)
        dens = densification_info if densification_info is not None and densification_info.numel() > 0 else None
        scratch = torch.empty(max(int(self.lib.fgs_shard_backward_scratch_bytes(n, k)), 1), dtype=torch.uint8, device=device)
        counts = (C.c_int32 * k)(*[int(x) for x in n_visible])
        self._check(self.lib.fgs_shard_backward(_ptr(acc_records) if total > 0 else None, counts, _ptr(primitive_buffer), _ptr(means),
                                                _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(sh_rest), _ptr(out[0]), _ptr(out[1]),
                                                _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), _ptr(out[5]), _ptr(dens), scratch.data_ptr(), n, k,
                                                S, _stream_of(device)), 'fgs_shard_backward')
        return out

    def shard_backward_adam_fused(self, acc_records: torch.Tensor, n_visible: Sequence[int], primitive_buffer: torch.Tensor, densification_info,
                                  params: Sequence[torch.Tensor], exp_avgs, exp_avg_sqs, views: Sequence[RasterizerSettings], step: int,
                                  lrs: Sequence[float], betas=(0.9, 0.999), eps: float = 1e-15) -> None:
        """`shard_backward` + Adam on the shard without materialising the gradients (<= 8 views). params / moments / lrs in
        optimizer-group order: means, sh0, sh_rest, opacities, scales, rotations."""
        device = self._check_params(tuple(params) + tuple(exp_avgs) + tuple(exp_avg_sqs), ['param/moment'] * 18)
        n, k = params[0].shape[0], len(views)
        total_rest = params[2].shape[1] if params[2].dim() == 3 else 0
        total = int(sum(n_visible))
        if len(n_visible) != k or (total > 0 and (acc_records.dtype != torch.float32 or acc_records.numel() < 9 * total
                                                  or not acc_records.is_contiguous() or acc_records.device != device)):
            raise RuntimeError('acc_records must be contiguous float32 [sum(n_visible), 9] and n_visible one entry per view')
        keep: list = []
        S = self._settings_array(views, total_rest, device, keep)
        dens = densification_info if densification_info is not None and densification_info.numel() > 0 else None
        scratch = torch.empty(max(int(self.lib.fgs_shard_backward_scratch_bytes(n, k)), 1), dtype=torch.uint8, device=device)
        counts = (C.c_int32 * k)(*[int(x) for x in n_visible])
        arr = lambda ts: (C.c_void_p * 6)(*[_ptr(t) for t in ts])
        self._check(self.lib.fgs_shard_backward_adam_fused(_ptr(acc_records) if total > 0 else None, counts, _ptr(primitive_buffer), arr(params),
                                                           arr(exp_avgs), arr(exp_avg_sqs), _ptr(dens), scratch.data_ptr(), n, k, S, int(step),
                                                           (C.c_double * 6)(*[float(x) for x in lrs]), float(betas[0]), float(betas[1]),
                                                           float(eps), _stream_of(device)), 'fgs_shard_backward_adam_fused')

    # -- the remaining exported operators (reference torch_bindings/filter3d.py, densification.py) ---------------------------
    def update_3d_filter(self, positions, w2c, filter_3d, visibility_mask, width, height, focal_x, focal_y, center_x, center_y,
                         near_plane, clipping_tolerance, distance2filter) -> None:
        device = self._check_params((positions, filter_3d), ('positions', 'filter_3d'))
        if visibility_mask.dtype != torch.bool or not visibility_mask.is_contiguous() or visibility_mask.device != device:
            raise RuntimeError('visibility_mask must be a contiguous bool tensor on the same device')
        w = w2c.to(device=device, dtype=torch.float32).contiguous()
        self._check(self.lib.fgs_update_3d_filter(_ptr(positions), w.data_ptr(), _ptr(filter_3d), _ptr(visibility_mask), positions.shape[0],
                                                  int(width), int(height), float(focal_x), float(focal_y), float(center_x), float(center_y),
                                                  float(near_plane), float(clipping_tolerance), float(distance2filter), _stream_of(device)),
                    'fgs_update_3d_filter')

    def relocation_adjustment(self, old_opacities, old_scales, n_samples_per_primitive):
        device = old_opacities.device
        op = old_opacities.to(torch.float32).contiguous()
        sc = old_scales.to(torch.float32).contiguous()
        ns = n_samples_per_primitive.to(device=device, dtype=torch.int64).contiguous()
        table = getattr(self, '_relocation_table', {}).get(device)
        if table is None:
            host = (C.c_float * 2500)()
            self._check(self.lib.fgs_relocation_table(host), 'fgs_relocation_table')
            table = torch.tensor(list(host), dtype=torch.float32, device=device)
            self._relocation_table = {**getattr(self, '_relocation_table', {}), device: table}
        n = op.shape[0]
        new_op = torch.empty((n, 1), dtype=torch.float32, device=device)
        new_sc = torch.empty((n, 3), dtype=torch.float32, device=device)
        self._check(self.lib.fgs_relocation_adjustment(_ptr(op), _ptr(sc), _ptr(ns), table.data_ptr(), _ptr(new_op), _ptr(new_sc), n,
                                                       _stream_of(device)), 'fgs_relocation_adjustment')
        return new_op, new_sc

    def add_noise(self, raw_scales, raw_rotations, raw_opacities, random_samples, means, current_lr: float) -> None:
        device = self._check_params((raw_scales, raw_rotations, raw_opacities, random_samples, means),
                                    ('raw_scales', 'raw_rotations', 'raw_opacities', 'random_samples', 'means'))
        self._check(self.lib.fgs_add_noise(_ptr(raw_scales), _ptr(raw_rotations), _ptr(raw_opacities), _ptr(random_samples), _ptr(means),
                                           means.shape[0], float(current_lr), _stream_of(device)), 'fgs_add_noise')

    # -- maintenance of the Gaussian set with the Adam moments (include/fgs_hip.h, "Next" row rank 1; Model.py:275-366, 459-463) -------
    def adaptive_density_control(self, densification_info, params: Sequence[torch.Tensor], exp_avgs, exp_avg_sqs, grad_threshold: float,
                                 min_opacity: float, prune_large_gaussians: bool, percent_dense: float, extent: float, noise_fn=None):
        """params (and optionally the two moment lists) in optimizer-group order: means, sh0, sh_rest, opacities, scales, rotations.
        Returns (new_params, new_exp_avgs or None, new_exp_avg_sqs or None, counts) with counts = (survivors, clones, children per
        copy, split). `noise_fn(n_rows)` supplies the [n_rows, 3] N(0,1) samples of the split (default torch.randn on the device)."""
        device = self._check_params(tuple(params) + (densification_info,), ['parameter'] * 6 + ['densification_info'])
        n = params[0].shape[0]
        total_rest = params[2].shape[1] if params[2].dim() == 3 else 0
        scratch = torch.empty(max(int(self.lib.fgs_adc_scratch_bytes(n)), 1), dtype=torch.uint8, device=device)
        counts = (C.c_int32 * 4)()
        self._check(self.lib.fgs_adc_plan(_ptr(densification_info), _ptr(params[4]), _ptr(params[5]), _ptr(params[3]), n, float(grad_threshold),
                                          float(min_opacity), int(bool(prune_large_gaussians)), float(percent_dense), float(extent),
                                          scratch.data_ptr(), counts, _stream_of(device)), 'fgs_adc_plan')
        kept, clones, children, split = (int(c) for c in counts)
        n_new = kept + clones + 2 * children
        noise = (noise_fn(2 * split) if noise_fn is not None else torch.randn((2 * split, 3), device=device)).to(device=device, dtype=torch.float32).contiguous()
        make = lambda t: torch.empty((n_new,) + tuple(t.shape[1:]), dtype=torch.float32, device=device)
        out_p = [make(t) for t in params]
        have_state = exp_avgs is not None
        if have_state:
            self._check_params(tuple(exp_avgs) + tuple(exp_avg_sqs), ['moment'] * 12)
        out_m = [make(t) for t in params] if have_state else None
        out_v = [make(t) for t in params] if have_state else None
        arr = lambda ts: (C.c_void_p * 6)(*[_ptr(t) for t in ts]) if ts is not None else None
        self._check(self.lib.fgs_adc_apply(arr(params), arr(exp_avgs), arr(exp_avg_sqs), arr(out_p), arr(out_m), arr(out_v), _ptr(noise),
                                           scratch.data_ptr(), n, total_rest, _stream_of(device)), 'fgs_adc_apply')
        return out_p, out_m, out_v, (kept, clones, children, split)

    def gather_rows(self, tensors: Sequence[torch.Tensor], index: torch.Tensor) -> list:
        """[t[index] for t in tensors] (float32, row-major) in one launch per 18 tensors: prune / sort of parameters and moments."""
        if not tensors:
            return []
        device = self._check_params(tuple(tensors), ['tensor'] * len(tensors))
        index = index.to(device=device, dtype=torch.int64).contiguous()
        rows = index.shape[0]
        outs = [torch.empty((rows,) + tuple(t.shape[1:]), dtype=torch.float32, device=device) for t in tensors]
        for first in range(0, len(tensors), 18):
            chunk, oc = tensors[first:first + 18], outs[first:first + 18]
            k = len(chunk)
            widths = [int(t.numel() // max(t.shape[0], 1)) if t.shape[0] > 0 else 0 for t in chunk]
            self._check(self.lib.fgs_gather_rows(k, (C.c_void_p * k)(*[_ptr(t) for t in chunk]), (C.c_void_p * k)(*[_ptr(t) for t in oc]),
                                                 (C.c_int32 * k)(*widths), _ptr(index), rows, _stream_of(device)), 'fgs_gather_rows')
        return outs

    def morton_order(self, means: torch.Tensor) -> torch.Tensor:
        """int64 permutation that sorts the points along a 30-bit Z-curve over their bounding box (Model.py:459-463), stable."""
        device = self._check_params((means,), ('means',))
        n = means.shape[0]
        order = torch.empty(n, dtype=torch.int64, device=device)
        if n == 0:
            return order
        lo, hi = means.min(dim=0).values.contiguous(), means.max(dim=0).values.contiguous()      # stay on the device: no host sync
        nbytes = int(self.lib.fgs_morton_order_temp_bytes(n))
        temp = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._check(self.lib.fgs_morton_order(_ptr(means), lo.data_ptr(), hi.data_ptr(), order.data_ptr(), n, temp.data_ptr(), nbytes,
                                              _stream_of(device)), 'fgs_morton_order')
        return order

    def profile_enable(self, enable, only: str | None = None) -> None:
        """enable=True: HIP events around every stage; only='adam': around that one stage (far less intrusive); False: off."""
        if enable and only is not None:
            names = list(self.profile_read().keys())
            self._check(self.lib.fgs_profile_enable(2 + names.index(only)), 'fgs_profile_enable')
        else:
            self.lib.fgs_profile_enable(int(bool(enable)))

    def profile_read(self) -> dict:
        """{stage: (total_ms, calls)} accumulated since the last read (HIP events on the launch stream)."""
        arr = (_lib.StageTime * 32)()
        k = self.lib.fgs_profile_read(arr, 32)
        return {arr[i].name.decode(): (arr[i].total_ms, arr[i].calls) for i in range(max(k, 0))}

    # -- introspection (tests / bench only) ------------------------------------------------------------------------------
    def blob_layout(self, which: int, n: int, width: int, height: int, n_instances: int, n_buckets: int) -> dict:
        entries = (_lib.BlobEntry * 32)()
        k = self.lib.fgs_blob_layout(which, n, width, height, n_instances, n_buckets, entries, 32)
        if k < 0:
            raise RuntimeError(self.lib.fgs_last_error().decode())
        return {entries[i].name.decode(): (entries[i].offset, entries[i].bytes) for i in range(k)}

    def view(self, blob: torch.Tensor, layout: dict, name: str, dtype: torch.dtype) -> torch.Tensor:
        off, nbytes = layout[name]
        return blob[off:off + nbytes].view(dtype)


_DEFAULT: Backend | None = None


def default_backend() -> Backend:
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Backend(_lib.library())
    return _DEFAULT
