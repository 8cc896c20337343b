"""View-parallel multi-GPU training step: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

The reference is single-GPU only (Renderer.py:58-61, no collective anywhere: SURVEY.md D4), so there is no reference
behaviour to match beyond "the update uses the sum of the per-view gradients". What shards naturally is the VIEW
(Trainer.py:180 draws one view per iteration); parameters are replicated, the only exchange is one gradient reduction per
optimizer step, and `densification_info` is summed when a densification step needs it.

Two exchange strategies over ONE contiguous fp32 arena holding all 59 floats per Gaussian (xGMI is point-to-point,
so few large collectives beat many small ones):

* mode='allreduce' : all-reduce the gradient arena, every rank runs the full Adam step (simplest; replicated state).
* mode='zero1'     : reduce-scatter the gradient arena -> each rank runs Adam on its 1/G slice of the arena (the dominant
                     per-iteration HBM term, 1 652 B/Gaussian, drops by G) -> all-gather the updated parameter slices.
                     Same wire volume as the all-reduce, 1/G of the optimizer traffic and state.

The rasterizer is reached through a `Backend` object (product: libfgs_hip.so; CPU tests: the simulation library), never
through autograd, so the gradient tensors are written straight into the arena.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist

from FasterGSCudaBackend._backend import Backend, RasterizerSettings

# arena segment order = optimizer group order (Model.py:238-245); backend.backward returns (means, scales, rotations,
# opacities, sh0, sh_rest)
SEGMENTS = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'opacities', 'scales', 'rotations')
_BACKWARD_ORDER = ('means', 'scales', 'rotations', 'opacities', 'sh_coefficients_0', 'sh_coefficients_rest')
_ALIGN = 64   # floats; keeps every segment and every rank slice 256-byte aligned


def l1_grad(image: torch.Tensor, target: torch.Tensor, scale: float) -> torch.Tensor:
    """d/dimage of scale * mean(|image - target|)."""
    return torch.sign(image - target) * (scale / image.numel())


class ViewParallelTrainer:
    def __init__(self, backend: Backend, params: dict, lrs: dict, *, mode: str = 'allreduce', group=None,
                 betas=(0.9, 0.999), eps: float = 1e-15, loss: str = 'l1_dssim', emulate_reduce_scatter: bool = False) -> None:
        assert mode in ('allreduce', 'zero1') and loss in ('l1', 'l1_dssim')
        self.be, self.mode, self.group, self.betas, self.eps, self.loss = backend, mode, group, betas, eps, loss
        self.emulate_reduce_scatter = bool(emulate_reduce_scatter)      # tests only: zero1's reduce-scatter as a full all-reduce
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        device = params['means'].device
        # ---- one arena for parameters, one for gradients ----
        self.layout, off = {}, 0
        for k in SEGMENTS:
            self.layout[k] = (off, params[k].numel(), tuple(params[k].shape))
            off += (params[k].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        chunk = (off + self.world - 1) // self.world
        self.chunk = (chunk + _ALIGN - 1) // _ALIGN * _ALIGN
        total = self.chunk * self.world
        self.param_arena = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad_arena = torch.zeros(total, dtype=torch.float32, device=device)
        self.params, self.grads = {}, {}
        for k in SEGMENTS:
            o, n, shape = self.layout[k]
            self.params[k] = self.param_arena[o:o + n].view(shape)
            self.params[k].copy_(params[k])
            self.grads[k] = self.grad_arena[o:o + n].view(shape)
        self.lrs = dict(lrs)
        self.step_count = 0
        self.n = params['means'].shape[0]
        self.densification_info = torch.zeros((2, self.n), dtype=torch.float32, device=device)
        # Adam state: full arena (allreduce) or this rank's slice only (zero1)
        state_len = total if mode == 'allreduce' else self.chunk
        self.exp_avg = torch.zeros(state_len, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(state_len, dtype=torch.float32, device=device)
        self.gather_from_copy = False
        if mode == 'zero1' and self.world > 1 and not self.emulate_reduce_scatter:
            self._probe_collectives(device)

    def _probe_collectives(self, device) -> None:
        """zero1 uses `reduce_scatter_tensor` with the output aliasing the input and an in-place `all_gather_into_tensor`. RCCL and the gloo of the
        installed torch do both; an older gloo may lack the first or mishandle the aliasing of the second (round-5 advisor finding). Tried ONCE here
        on 4 floats per rank with known answers; every rank takes the same decision (MIN over ranks): all-reduce form of the reduce-scatter, a
        copied all-gather input."""
        w, r, expect = self.world, self.rank, float(self.world * (self.world + 1) // 2)
        ok = torch.ones(2, dtype=torch.float32, device=device)
        try:
            buf = torch.full((4 * w,), float(r + 1), dtype=torch.float32, device=device)
            mine = buf[4 * r:4 * r + 4]
            dist.reduce_scatter_tensor(mine, buf, group=self.group)
            ok[0] = float(bool((mine == expect).all()))
        except (RuntimeError, NotImplementedError, AttributeError):
            ok[0] = 0.0
        try:
            buf = torch.zeros(4 * w, dtype=torch.float32, device=device)
            buf[4 * r:4 * r + 4] = float(r + 1)
            dist.all_gather_into_tensor(buf, buf[4 * r:4 * r + 4], group=self.group)
            ok[1] = float(bool((buf.view(w, 4) == torch.arange(1, w + 1, dtype=torch.float32, device=device)[:, None]).all()))
        except (RuntimeError, NotImplementedError, AttributeError):
            ok[1] = 0.0
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        self.emulate_reduce_scatter = not bool(ok[0])
        self.gather_from_copy = not bool(ok[1])

    # ---- the pieces of one step -----------------------------------------------------------------------------------
    def _render_backward(self, settings: RasterizerSettings, grad_fn: Callable[[torch.Tensor], torch.Tensor], update_densification: bool):
        p = self.params
        res = self.be.forward(p['means'], p['scales'], p['rotations'], p['opacities'], p['sh_coefficients_0'], p['sh_coefficients_rest'], settings)
        grad_image = grad_fn(res.image)
        self.be.backward(self.densification_info if update_densification else None, grad_image, res.image, p['means'], p['scales'],
                         p['rotations'], p['opacities'], p['sh_coefficients_rest'], res.buffers, settings, res.state,
                         out=tuple(self.grads[k] for k in _BACKWARD_ORDER))
        return res.image

    def _segments_in(self, lo: int, hi: int):
        """Arena segments intersected with [lo, hi): (name, start, stop) in arena coordinates."""
        out = []
        for k in SEGMENTS:
            o, n, _ = self.layout[k]
            a, b = max(o, lo), min(o + n, hi)
            if a < b:
                out.append((k, a, b))
        return out

    def _adam(self, lo: int, hi: int, state_offset: int) -> None:
        segs = self._segments_in(lo, hi)
        if not segs:
            return
        g = [self.grad_arena[a:b] for _, a, b in segs]
        p = [self.param_arena[a:b] for _, a, b in segs]
        m = [self.exp_avg[a - state_offset:b - state_offset] for _, a, b in segs]
        v = [self.exp_avg_sq[a - state_offset:b - state_offset] for _, a, b in segs]
        self.be.adam_step_multi(g, p, m, v, [self.step_count] * len(segs), [self.lrs[k] for k, _, _ in segs], self.betas[0], self.betas[1], self.eps)

    def _reduce_scatter(self) -> torch.Tensor:
        """This rank's slice of the summed gradient arena, in place (output = a view of the input: the in-place form of the collective). The SAME
        call on every backend -- RCCL on the GPUs, gloo in the CPU tests (torch's gloo backend implements reduce_scatter_tensor) -- so the CPU tests
        execute the call path the hardware run takes. `emulate_reduce_scatter` (tests only) forces the all-reduce form a backend without the
        collective would need; the two are compared in tests/test_distributed.py."""
        mine = self.grad_arena[self.rank * self.chunk:(self.rank + 1) * self.chunk]
        if self.emulate_reduce_scatter:
            dist.all_reduce(self.grad_arena, group=self.group)
        else:
            dist.reduce_scatter_tensor(mine, self.grad_arena, group=self.group)
        return mine

    def image_gradient(self, image: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """d/dimage of (1/world) * loss(image, target); loss = 0.8 L1 + 0.2 DSSIM (Trainer.py:52-53) or plain L1."""
        if self.loss == 'l1':
            return l1_grad(image, target, 1.0 / self.world)
        grad = self.be.l1_dssim(image, target, 0.8, 0.2, with_grad=True)[1]
        return grad if self.world == 1 else grad * (1.0 / self.world)

    # ---- public ---------------------------------------------------------------------------------------------------
    def step(self, settings: RasterizerSettings, target: torch.Tensor, *, update_densification: bool = True) -> torch.Tensor:
        """One optimizer step over a global batch of `world` views (this rank's view = `settings`). The loss is the mean
        over ranks of the per-view L1, so the exchanged quantity is the plain SUM of per-rank gradients."""
        self.step_count += 1
        image = self._render_backward(settings, lambda img: self.image_gradient(img, target), update_densification)
        if not dist.is_initialized():
            self._adam(0, self.param_arena.numel(), 0)
        elif self.mode == 'allreduce':
            dist.all_reduce(self.grad_arena, group=self.group)
            self._adam(0, self.param_arena.numel(), 0)
        else:
            self._reduce_scatter()
            lo = self.rank * self.chunk
            self._adam(lo, lo + self.chunk, lo)
            # in place: the input is this rank's slice of the output arena (the in-place form of all-gather; no 1/G-arena copy per step)
            mine = self.param_arena[lo:lo + self.chunk]
            dist.all_gather_into_tensor(self.param_arena, mine.clone() if self.gather_from_copy else mine, group=self.group)
        return image

    def gather_densification_info(self) -> torch.Tensor:
        """Sum of the per-rank statistics; every rank then takes identical densify / prune decisions (Model.py:312-366).
        Returns a NEW tensor: the local accumulator keeps only this rank's statistics, so calling this twice before a reset
        does not count earlier steps G times."""
        total = self.densification_info.clone()
        if self.world > 1:
            dist.all_reduce(total, group=self.group)
        return total

    def set_learning_rates(self, lrs: dict) -> None:
        self.lrs.update(lrs)
