"""Operator surface of the rasterizer: diff_rasterize / rasterize / RasterizerSettings.

Names, argument order and semantics follow the reference's torch_bindings/rasterization.py:41-156 so that
Renderer.py:72-123 runs unmodified against this package; the implementation underneath is libfgs_hip.so through ctypes.
"""
from __future__ import annotations

from typing import Any

import torch
from torch.autograd.function import once_differentiable

from ._backend import RasterizerSettings, default_backend


_GRAD_OUT = None   # optional provider of preallocated gradient tensors (harness/distributed.py packs them into one arena)


def set_gradient_buffers(provider) -> None:
    """provider() -> six tensors (means, scales, rotations, opacities, sh0, sh_rest order) or None; None resets."""
    global _GRAD_OUT
    _GRAD_OUT = provider


# ---- host-synchronisation-free training forward (fgs_forward_async) ------------------------------------------------------------------
# The reference blocks the host three times per forward pass (forward.cu:100,102,234) to size its buffers; fgs_forward once. With
# `set_async_forward(True)` the training path sizes the instance stages from what an earlier pass OF THE SAME VIEW needed -- that view's
# instances-per-Gaussian ratio x the current Gaussian count x `headroom` -- and reads the counts back asynchronously; they are looked at in
# `backward` (by then the copy has long completed). A view that has not been seen yet, and any pass that does not need gradients
# (torch.no_grad(), detached inputs: nobody would ever look at the counts), takes the synchronous path -- a ratio borrowed from other views
# overflows in the first epoch (round-2 advisor finding). A view is recognised by the IDENTITY of its w2c tensor (address + a weak reference to
# the tensor object + its version counter, plus image size and intrinsics): an address the caching allocator hands to another tensor, or a w2c
# edited in place, is a new view, never a cache hit (round-3 advisor finding); a training loop that builds a fresh w2c every iteration therefore
# stays synchronous -- correct, only without the saving. The table is bounded (least recently used views are dropped).
# If a pass still needs more than its capacity (the instance count of one view jumped by more than the headroom between two of its visits), its
# image and therefore the loss gradient were incomplete: backward reports it (RuntimeWarning), refreshes the view's ratio, returns ZERO gradients
# (written into the caller's gradient buffers when a provider is set; densification_info is left alone) and marks the step invalid:
# `FusedAdam.step` consumes that mark and skips the step -- no moment decay, no step count, no parameter motion on momentum (any other optimizer
# can ask `take_async_overflow()`). A pass that asked for gradients but whose backward never runs (an evaluation loop without no_grad) is checked
# lazily by the next forward pass, which warns if that image was truncated.
# Scopes: the per-view table records instances PER GAUSSIAN, a property of the model as much as of the view. A process that trains two models over
# the same view tensors gives each its own table with `async_forward_scope(name)` (a context manager around that model's iterations); everything
# outside such a block uses the default scope. Switches (`set_async_forward`) and counters are per scope as well.
def _new_async_table() -> dict:
    return {'enabled': False, 'headroom': 1.25, 'ratio': 0.0, 'overflows': 0, 'per_view': {}, 'pending': {}, 'step_invalid': False, 'ticket': 0}


_ASYNC_SCOPES = {None: _new_async_table()}
_ASYNC = _ASYNC_SCOPES[None]          # the ACTIVE scope's table (rebound by async_forward_scope)
_ASYNC_MAX_VIEWS = 1024


class async_forward_scope:
    """`with async_forward_scope('model_b'): ...` -- the asynchronous-forward bookkeeping (switch, per-view ratios, pending checks, overflow mark) of
    the block is kept apart from every other scope's. Scopes persist across blocks of the same name; `async_forward_scope.drop(name)` forgets one."""

    def __init__(self, name):
        self.name, self._outer = name, None

    def __enter__(self):
        global _ASYNC
        self._outer = _ASYNC
        _ASYNC = _ASYNC_SCOPES.setdefault(self.name, _new_async_table())
        return self

    def __exit__(self, *exc):
        global _ASYNC
        _ASYNC = self._outer

    @staticmethod
    def drop(name) -> None:
        if name is not None:
            _ASYNC_SCOPES.pop(name, None)

# ---- live-block hand-over from this backward pass to FusedAdam.step ------------------------------------------------------------------
# One third of the Gaussians is invisible in a view; their gradients are zeros that the backward pass writes (the gradient tensors are dense and
# valid for any reader) and the optimizer reads back. The backward pass also leaves one byte per block of 64 Gaussians ("any visible"), and
# FusedAdam.step skips READING the gradients of dead blocks -- if, and only if, the gradients it is handed are still exactly what this
# backward pass wrote. That is established without trusting anybody: the six gradients are views into ONE arena that this registry keeps alive
# (their addresses cannot be re-used by another tensor while registered). The registry holds the ARENA and the views' addresses / shapes, never
# the view tensors themselves: autograd adopts an incoming gradient as `.grad` only if nobody else references that tensor (it looks at the
# reference count of the view, not of its storage) and clones it otherwise -- 708 MB per iteration at 3 M Gaussians. A match needs the same address, the same shape and an unchanged version counter (views share the arena's: any
# in-place edit -- accumulation of a second backward, clipping, scaling -- shows). Anything else takes the ordinary path; results are
# bit-identical either way. Writes the version counter does not see (`.grad.data.add_(...)`, raw pointers) are the kernel's business: it reads
# one sentinel float per dead block and tensor and, unless that is +-0, the block's gradients after all (csrc/preprocess_backward.hip).
# The registry holds ONE registration per model, keyed by the address of the `means` tensor of the pass (two models in one process do not evict
# each other's registration; a model's next backward pass replaces its own). An optimizer names the parameters it owns when it asks.
_LIVE = {'enabled': True, 'slots': {}, 'matched': 0, 'missed': 0}
_LIVE_MAX_SLOTS = 8
_ALIGN_FLOATS = 64          # every gradient starts on a 256-byte boundary (16-byte loads in the optimizer kernel)


def set_live_block_handover(enabled: bool) -> None:
    _LIVE['enabled'] = bool(enabled)
    clear_live_blocks()


def live_block_stats() -> dict:
    return {'matched': _LIVE['matched'], 'missed': _LIVE['missed']}


def clear_live_blocks(owned=None) -> None:
    """Forgets the registrations of the passes over the given parameter tensors (an optimizer's own), or all of them."""
    if owned is None:
        _LIVE['slots'].clear()
        return
    for t in owned:
        _LIVE['slots'].pop(t.data_ptr(), None)


def _gradient_arena(shapes, device):
    offsets, total = [], 0
    for sh in shapes:
        offsets.append(total)
        numel = 1
        for d in sh:
            numel *= d
        total += (numel + _ALIGN_FLOATS - 1) // _ALIGN_FLOATS * _ALIGN_FLOATS
    arena = torch.empty(max(total, 1), dtype=torch.float32, device=device)
    views = []
    for sh, off in zip(shapes, offsets):
        numel = 1
        for d in sh:
            numel *= d
        views.append(arena[off:off + numel].view(sh))
    return arena, tuple(views)


def match_live_blocks(gradients, owned=None) -> 'torch.Tensor | None':
    """The flags of a registered backward pass if `gradients` (the tensors an optimizer is about to consume, one per parameter group) are
    exactly the tensors it wrote -- same addresses, shapes, untouched since -- else None. `owned`: the optimizer's parameter tensors; only the
    registration of a pass over one of them is looked at (and consumed either way). Without it every registration is a candidate."""
    slots = _LIVE['slots']
    keys = list(slots) if owned is None else [t.data_ptr() for t in owned if t.data_ptr() in slots]
    if not keys or len(gradients) == 0:
        return None
    first = gradients[0].data_ptr()
    key = next((k for k in keys if any(address == first for address, _ in slots[k]['views'])), keys[0])
    slot = slots.pop(key)
    arena, flags, views, version = slot['arena'], slot['flags'], slot['views'], slot['version']
    by_address = {address: shape for address, shape in views if address != 0}
    seen = set()
    for g in gradients:
        shape = by_address.get(g.data_ptr())
        if (shape is None or g.data_ptr() in seen or tuple(g.shape) != shape or g.dtype != torch.float32 or not g.is_contiguous()
                or g.device != arena.device or g._version != version):
            _LIVE['missed'] += 1
            return None
        seen.add(g.data_ptr())
    _LIVE['matched'] += 1
    return flags


def set_async_forward(enabled: bool, headroom: float = 1.25) -> None:
    _ASYNC.update(enabled=bool(enabled), headroom=float(headroom))
    if not enabled:
        _ASYNC.update(ratio=0.0, per_view={}, pending={}, step_invalid=False)


def async_forward_stats() -> dict:
    """'ratio': the largest instances-per-Gaussian ratio of any view so far (informational); 'views': views with a recorded ratio."""
    return {'enabled': _ASYNC['enabled'], 'headroom': _ASYNC['headroom'], 'ratio': _ASYNC['ratio'], 'overflows': _ASYNC['overflows'],
            'views': len(_ASYNC['per_view']), 'unchecked_passes': len(_ASYNC['pending'])}


def take_async_overflow(owned=None) -> bool:
    """True once after a backward pass had to return zero gradients because its forward pass overflowed its capacity: the optimizer step that
    would consume them must be skipped (FusedAdam.step does). The mark names the parameter tensors of that pass (their storage addresses): with
    `owned` (an iterable of tensors) only an optimizer that owns one of them takes it -- a second FusedAdam over other parameters neither
    consumes the mark nor skips its own step; without arguments any pending mark is taken. The next forward pass over the same parameters
    drops a mark nobody took (no optimizer step followed: another optimizer, an exception), so it cannot skip a later, valid step."""
    if owned is None:
        marked, _ASYNC['step_invalid'] = _ASYNC['step_invalid'], False
        return bool(marked)
    mine = {a for t in owned for a in _addresses(t)}
    for table in _ASYNC_SCOPES.values():          # the optimizer may step outside the scope block its passes ran in
        if table['step_invalid'] and not mine.isdisjoint(table['step_invalid']):
            table['step_invalid'] = False
            return True
    return False


def _addresses(t: torch.Tensor) -> tuple:
    """What identifies `t` in an overflow mark: its own address and the address of its storage -- a `.contiguous()` copy is neither, but a view, a
    slice or a reshaped alias of a marked parameter (or a parameter that is a view into a marked arena) still is."""
    if t.numel() == 0:
        return ()
    return (t.data_ptr(), t.untyped_storage().data_ptr())


def _tensor_version(t: torch.Tensor) -> int:
    """The autograd version counter, or None for tensors created under torch.inference_mode() (they have none, reading it raises): such a view
    is never cached -- every pass over it is a synchronous one."""
    try:
        return t._version
    except RuntimeError:
        return None


def _view_key(settings: RasterizerSettings):
    return (settings.w2c.data_ptr(), int(settings.width), int(settings.height), float(settings.focal_x), float(settings.focal_y),
            float(settings.center_x), float(settings.center_y))


def _view_ratio(A: dict, key, w2c: torch.Tensor) -> float:
    """The recorded instances-per-Gaussian ratio of this view, or 0.0 if `w2c` is not the very tensor (object and content version) it was recorded for."""
    entry = A['per_view'].get(key)
    if entry is None:
        return 0.0
    ratio, ref, version = entry
    if ref() is not w2c or version is None or _tensor_version(w2c) != version:
        del A['per_view'][key]
        return 0.0
    A['per_view'][key] = A['per_view'].pop(key)          # most recently used last
    return ratio


def _record_ratio(A: dict, key, w2c: torch.Tensor, ratio: float) -> None:
    import weakref
    table = A['per_view']
    table.pop(key, None)
    table[key] = (ratio, weakref.ref(w2c), _tensor_version(w2c))
    while len(table) > _ASYNC_MAX_VIEWS:
        del table[next(iter(table))]
    A['ratio'] = max(A['ratio'], ratio)


def _check_abandoned_passes(A: dict) -> None:
    """Asynchronous passes whose backward never ran: look at their counts now (their copies completed long ago) and say so if an image was truncated."""
    pending = A['pending']
    for ticket in list(pending):
        host, event, capacity = pending[ticket]
        if event is not None and not event.query():
            continue
        del pending[ticket]
        if int(host[2]) != 0:
            import warnings
            A['overflows'] += 1
            warnings.warn(f'FasterGS async forward: an earlier pass whose backward never ran needed {int(host[1])} instances but had capacity {capacity}: '
                          f'the image it returned was incomplete (render evaluation views under torch.no_grad(): those passes are sized synchronously)',
                          RuntimeWarning)
    while len(pending) > 64:                      # never unbounded: drop the oldest tickets unchecked
        del pending[next(iter(pending))]


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        # Renderer.py:58-59 raises in CPU mode as well; there is no CPU implementation behind this package
        raise RuntimeError('FasterGS rasterizer: tensors must live on a ROCm/HIP device (no CPU implementation)')


class _Rasterize(torch.autograd.Function):
    """Differentiable w.r.t. the six parameter tensors (rasterization.py:41-110 of the reference)."""

    @staticmethod
    def forward(ctx: Any, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, densification_info,
                rasterizer_settings: RasterizerSettings) -> torch.Tensor:
        _require_gpu(means)
        be, n, A = default_backend(), means.shape[0], _ASYNC          # the active scope's table; backward uses the same one
        capacity, key = None, None
        if A['step_invalid']:
            # the step of that overflowed pass never came (no optimizer took the mark: another optimizer class, an exception, inputs that were
            # copies of the optimizer's tensors): a stale mark must not skip a later, valid step -- whatever tensors this pass is over
            A['step_invalid'] = False
        if A['enabled'] and n > 0:
            if A['pending']:
                _check_abandoned_passes(A)
            key = _view_key(rasterizer_settings)
            ratio = _view_ratio(A, key, rasterizer_settings.w2c)
            # only a pass whose backward will run ever looks at the asynchronous counts; everything else is checked now (synchronously)
            if ratio > 0.0 and any(ctx.needs_input_grad[:6]):
                capacity = int(ratio * n * A['headroom']) + 4096
        res = be.forward(means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, rasterizer_settings, capacity)
        if capacity is None:
            if key is not None:
                _record_ratio(A, key, rasterizer_settings.w2c, res.state[1] / n)
            ctx.async_check = None
        else:
            host, event = be.forward_counts(res, n)
            A['ticket'] += 1
            A['pending'][A['ticket']] = (host, event, capacity)      # consumed by backward; looked at by a later forward otherwise
            ctx.async_check = (host, event, key, A['ticket'])
        ctx.rasterizer_settings = rasterizer_settings
        ctx.buffer_state = res.state
        ctx.sh0_addresses = _addresses(sh_coefficients_0)
        ctx.async_table = A
        ctx.save_for_backward(res.image, means, scales, rotations, opacities, sh_coefficients_rest, *res.buffers)
        ctx.densification_info = densification_info
        ctx.mark_non_differentiable(densification_info)
        return res.image

    @staticmethod
    @once_differentiable
    def backward(ctx: Any, grad_image: torch.Tensor):
        image, means, scales, rotations, opacities, sh_rest, *buffers = ctx.saved_tensors
        state, A = ctx.buffer_state, ctx.async_table
        if ctx.async_check is not None:
            host, event, key, ticket = ctx.async_check
            A['pending'].pop(ticket, None)
            if event is not None:
                event.synchronize()
            n = means.shape[0]
            _record_ratio(A, key, ctx.rasterizer_settings.w2c, int(host[1]) / max(n, 1))
            if int(host[2]) != 0:          # the capacity was too small: the image (and the loss gradient) missed the instances beyond it
                import warnings
                A['overflows'] += 1
                warnings.warn(f'FasterGS async forward: {int(host[1])} instances exceeded the capacity {state[1]} of this pass: its image was '
                              f'incomplete, so this backward pass returns zero gradients and FusedAdam.step skips the step (the view is rendered with the '
                              f'right capacity next time)', RuntimeWarning)
                clear_live_blocks([means])
                A['step_invalid'] = frozenset(a for t in (means, scales, rotations, opacities, sh_rest) for a in _addresses(t)) | frozenset(ctx.sh0_addresses)
                if _GRAD_OUT is not None:          # a consumer that reads the provider's arena directly must not see the previous step's gradients
                    zeros = tuple(_GRAD_OUT())
                    for z in zeros:
                        z.zero_()
                else:
                    total_rest = sh_rest.shape[1] if sh_rest.dim() == 3 else 0
                    zeros = tuple(torch.zeros(sh, dtype=torch.float32, device=means.device)
                                  for sh in ((n, 3), (n, 3), (n, 4), (n, 1), (n, 1, 3), (n, total_rest, 3)))
                return (*zeros, None, None)
        n = means.shape[0]
        if _LIVE['enabled'] and _GRAD_OUT is None and n > 0:
            total_rest = sh_rest.shape[1] if sh_rest.dim() == 3 else 0
            arena, views = _gradient_arena(((n, 3), (n, 3), (n, 4), (n, 1), (n, 1, 3), (n, total_rest, 3)), means.device)
            flags = torch.empty((n + 63) // 64, dtype=torch.uint8, device=means.device)
            grads = default_backend().backward(ctx.densification_info, grad_image, image, means, scales, rotations, opacities, sh_rest,
                                               buffers, ctx.rasterizer_settings, state, out=views, live_blocks=flags)
            slots = _LIVE['slots']
            slots.pop(means.data_ptr(), None)
            slots[means.data_ptr()] = {'arena': arena, 'version': arena._version, 'flags': flags,
                                       'views': tuple((v.data_ptr() if v.numel() else 0, tuple(v.shape)) for v in views)}
            while len(slots) > _LIVE_MAX_SLOTS:          # models that backpropagate but never step: their arenas are not kept alive for ever
                del slots[next(iter(slots))]
            del views
        else:
            clear_live_blocks([means])
            grads = default_backend().backward(ctx.densification_info, grad_image, image, means, scales, rotations, opacities, sh_rest,
                                               buffers, ctx.rasterizer_settings, state,
                                               out=_GRAD_OUT() if _GRAD_OUT is not None else None)
        return (*grads, None, None)   # densification_info, rasterizer_settings


def diff_rasterize(means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, opacities: torch.Tensor,
                   sh_coefficients_0: torch.Tensor, sh_coefficients_rest: torch.Tensor, densification_info: torch.Tensor,
                   rasterizer_settings: RasterizerSettings) -> torch.Tensor:
    """Training render: image [3,H,W]; densification_info [2,N] is updated in backward, or pass torch.empty(0)."""
    return _Rasterize.apply(means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, densification_info,
                            rasterizer_settings)


def rasterize(means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, opacities: torch.Tensor,
              sh_coefficients_0: torch.Tensor, sh_coefficients_rest: torch.Tensor, rasterizer_settings: RasterizerSettings,
              to_chw: bool, clamp_output: bool = True) -> torch.Tensor:
    """Forward-only render (the reference's benchmark path, Renderer.py:107-123): [3,H,W] or [H,W,3]."""
    _require_gpu(means)
    return default_backend().inference(means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest,
                                       rasterizer_settings, to_chw, clamp_output)


def update_pruning_scores(scores: torch.Tensor, means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor,
                          opacities: torch.Tensor, sh_coefficients_0: torch.Tensor, sh_coefficients_rest: torch.Tensor,
                          rasterizer_settings: RasterizerSettings) -> None:
    """Speedy-Splat importance scores of one view, accumulated into `scores` (reference rasterization.py:159-178,
    called per training view from Renderer.py:141-156)."""
    _require_gpu(means)
    default_backend().pruning_scores(scores, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest,
                                     rasterizer_settings)
