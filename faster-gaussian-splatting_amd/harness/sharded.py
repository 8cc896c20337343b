"""Gaussian-sharded view-parallel training step: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

`ViewParallelTrainer` (distributed.py) replicates the parameters, so every step moves the whole gradient / parameter set
over xGMI: 2 x 236 B per Gaussian per rank (reduce-scatter + all-gather), 708 MB each way at 3 M Gaussians. xGMI is
point-to-point (7 links x ~77 GB/s per direction per GPU; TWO ranks share ONE link), so that exchange costs more than the
whole single-GPU iteration at G = 2 and still ~3 ms at G = 8.

Here rank r OWNS the Gaussians r, r+G, r+2G, ... (parameters, gradients, Adam moments: 1/G of the memory) and the step is
cut where the per-Gaussian data is small (include/fgs_hip.h, "Gaussian-sharded multi-GPU path"):

    every rank, the G views of the step in one launch: K1 on its shard          -> 56-B records of the visible
    all-to-all #1                                       records of view v        -> rank v
    rank v:                                             K2..K10, loss, K11       -> 36-B accumulator per record
    all-to-all #2                                       accumulators             -> back to the owners
    every rank, all views in one launch:                K12 on its shard (sums over views), then ONE Adam launch on the shard

Wire volume per rank and step: (56 + 36) B x V x (G-1)/G  (V = visible Gaussians of a view, ~2 M at S2) = ~160 MB at G = 8
instead of 1 240 MB, spread over all 7 links; K1 and K12 do the same total work as on one GPU, Adam does 1/G of it.
One host synchronisation per step (the G x G table of record counts).

The per-view loss is averaged over the G views, so the update equals one Adam step on the mean of the per-view losses --
the same mathematics as `ViewParallelTrainer`, up to fp32 summation order.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist

from FasterGSCudaBackend import _lib
from FasterGSCudaBackend._backend import Backend, RasterizerSettings

from .distributed import SEGMENTS, _ALIGN, _BACKWARD_ORDER, l1_grad


INTERLEAVE_SHARDS = True      # A/B switch (tools/archive/ab_sharded_order.py, tests): False = the renderer keeps the records in the order they arrive


def shard_of(params: dict, rank: int, world: int) -> dict:
    """Strided ownership: every shard sees the same spatial distribution, so the per-(shard, view) record counts -- the
    all-to-all message sizes -- are balanced whatever the memory order (Morton, Model.py:357-366) of the scene."""
    return {k: v[rank::world].contiguous() for k, v in params.items()}


def _merge_shards(parts: Sequence[torch.Tensor], strided: bool) -> torch.Tensor:
    """Shards back into one tensor. While every shard is still the strided slice it was created from (rank r holds rows
    r, r+G, ...) the rows are interleaved back into the global order; once any owner has densified / pruned / re-ordered its
    shard (`rebuild_from`) there is no global order left to restore and the shards are concatenated in rank order (callers that
    want locality Morton-sort the result)."""
    world, total = len(parts), sum(int(p.shape[0]) for p in parts)
    if strided and all(int(p.shape[0]) == len(range(s, total, world)) for s, p in enumerate(parts)):
        full = torch.empty((total,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
        for s, p in enumerate(parts):
            full[s::world] = p
        return full
    return torch.cat(list(parts), dim=0)


class ShardedTrainer:
    def __init__(self, backend: Backend, shard_params: dict, lrs: dict, *, group=None, betas=(0.9, 0.999), eps: float = 1e-15,
                 loss: str = 'l1_dssim', fused: bool = True, extent: float = 1.0, lr_config: dict | None = None,
                 max_sh_degree: int | None = None) -> None:
        assert loss in ('l1', 'l1_dssim')
        # what owner-side maintenance (as_gaussians -> harness.densify) needs to decide exactly like the single-GPU model:
        # the training cameras' extent (Model.py:312-366: percent_dense * extent, prune-large threshold), the learning-rate
        # configuration and the SH degree cap
        self.extent, self.lr_config, self.max_sh_degree = float(extent), lr_config, max_sh_degree
        self.be, self.group, self.betas, self.eps, self.loss = backend, group, betas, eps, loss
        self.fused = fused            # phase C as ONE fused K12 + Adam pass (gradients never materialised) when the batch has <= 8 views
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        device = shard_params['means'].device
        self.device = device
        self.n = shard_params['means'].shape[0]
        self.total_sh_rest = shard_params['sh_coefficients_rest'].shape[1]
        # one arena each for the shard's parameters, gradients and the two Adam moments -> ONE Adam launch
        self.layout, off = {}, 0
        for k in SEGMENTS:
            self.layout[k] = (off, shard_params[k].numel(), tuple(shard_params[k].shape))
            off += (shard_params[k].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.param_arena = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad_arena = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        self.params, self.grads = {}, {}
        for k in SEGMENTS:
            o, n, shape = self.layout[k]
            self.params[k] = self.param_arena[o:o + n].view(shape)
            self.params[k].copy_(shard_params[k])
            self.grads[k] = self.grad_arena[o:o + n].view(shape)
        self.lrs = dict(lrs)
        self.step_count = 0
        self.strided = True           # still the slice rank::world of the scene it was created from (until rebuild_from)
        self.densification_info = torch.zeros((2, self.n), dtype=torch.float32, device=device)
        # send side of exchange #1: the shard's records for each view of the step, and their (V, I) counts
        self.records = torch.empty((self.world, max(self.n, 1), _lib.SPLAT_RECORD_BYTES), dtype=torch.uint8, device=device)
        self.counts = torch.zeros((self.world, 2), dtype=torch.int32, device=device)
        self.last_counts = None       # [G shards, G views, 2] of the last step (host), for reporting
        # Exposed communication time (bench.py --gpus N): the three exchanges of a step are bracketed by events when `time_comm` is set.
        # Nothing overlaps them: a step is a chain (K1 -> counts -> records -> K2..K11 -> accumulators -> K12 + Adam), every phase needs
        # ALL of the previous one's output (K12 sums a Gaussian's accumulators over the G views in registers), and the next step's K1 needs
        # this step's parameter update -- so exposed = total.
        self.time_comm, self._comm_events = False, []

    # ---- exchanges ------------------------------------------------------------------------------------------------
    def _timed(self, fn):
        if not (self.time_comm and self.device.type == 'cuda'):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self._comm_events.append((a, b))
        return out

    def comm_ms_per_step(self) -> float:
        """Mean time per step between the start and the end of the step's exchanges on this rank's stream (set time_comm = True first);
        clears the record. Includes the wait for the slowest peer, as an exposed exchange does."""
        if not self._comm_events or self.step_count == 0:
            return 0.0
        torch.cuda.synchronize(self.device)
        total = sum(a.elapsed_time(b) for a, b in self._comm_events)
        steps = max(len(self._comm_events) // 3, 1)              # three exchanges per step
        self._comm_events = []
        return total / steps

    def _all_to_all(self, pieces: Sequence[torch.Tensor], recv_rows: Sequence[int]) -> torch.Tensor:
        """pieces[j] goes to rank j; returns the concatenation of what ranks 0..G-1 sent here (rows along dim 0)."""
        if self.world == 1:
            return pieces[0]
        # an uneven all_to_all_single with wrong split lists fails deep inside the backend (or hangs a peer): check them here, readably
        if len(pieces) != self.world or len(recv_rows) != self.world or any(int(x) < 0 for x in recv_rows):
            raise ValueError(f'rank {self.rank}: all-to-all over {self.world} ranks got {len(pieces)} pieces to send and {len(recv_rows)} receive counts '
                             f'{[int(x) for x in recv_rows]}: one (non-negative) entry per rank is required')

        def run():
            send = torch.cat(list(pieces), dim=0)
            recv = torch.empty((sum(recv_rows),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
            dist.all_to_all_single(recv, send, [int(x) for x in recv_rows], [int(p.shape[0]) for p in pieces], group=self.group)
            return recv
        return self._timed(run)

    def _gather_counts(self) -> torch.Tensor:
        if self.world == 1:
            return self.counts.cpu().view(1, 1, 2)

        def run():
            table = torch.empty(self.world * self.world * 2, dtype=torch.int32, device=self.device)
            dist.all_gather_into_tensor(table, self.counts.view(-1), group=self.group)
            return table
        return self._timed(run).cpu().view(self.world, self.world, 2)       # the one host synchronisation of the step

    def image_gradient(self, image: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if self.loss == 'l1':
            return l1_grad(image, target, 1.0 / self.world)
        grad = self.be.l1_dssim(image, target, 0.8, 0.2, with_grad=True)[1]
        return grad if self.world == 1 else grad * (1.0 / self.world)

    # ---- the three compute phases of a step (no communication inside) ------------------------------------------------------
    def project(self, views: Sequence[RasterizerSettings]) -> torch.Tensor:
        """Phase A (owner): K1 on the shard for every view (one batched launch); fills self.records / self.counts, returns the
        primitive buffer of the step."""
        p = self.params
        tensors = (p['means'], p['scales'], p['rotations'], p['opacities'], p['sh_coefficients_0'], p['sh_coefficients_rest'])
        return self.be.shard_preprocess(*tensors, views, self.records, self.counts)

    def render(self, records: torch.Tensor, n_instances: int, view: RasterizerSettings, target: torch.Tensor, shard_counts: Sequence[int] | None = None):
        """Phase B (renderer of `view`): K2..K10, loss gradient, K11 -> (image, accumulator records [n_records, 9]). `shard_counts`: how many of the
        records each shard sent (in concatenation order): the renderer interleaves them, which gives K11 back the locality of the scene's Morton
        order (profiles/r04_ab_sharded_order.txt); the accumulator records still come out in the order the records came in."""
        if not INTERLEAVE_SHARDS:
            shard_counts = None
        res = self.be.forward_from_records(records.view(-1), records.shape[0], n_instances, view, self.total_sh_rest, shard_counts=shard_counts)
        grad_image = self.image_gradient(res.image, target)
        acc = self.be.backward_to_records(grad_image, res.image, res.buffers, view, res.state, self.total_sh_rest, shard_counts=shard_counts)
        return res.image, acc

    def finish(self, views: Sequence[RasterizerSettings], prim: torch.Tensor, acc_back: torch.Tensor, sent: Sequence[int],
               update_densification: bool) -> None:
        """Phase C (owner): K12 on the shard, gradients summed over the views in registers (one launch), then one Adam launch."""
        p = self.params
        dens = self.densification_info if update_densification else None
        self.step_count += 1
        if self.fused and len(views) <= 8:
            view_of = lambda arena, k: arena[self.layout[k][0]:self.layout[k][0] + self.layout[k][1]].view(self.layout[k][2])
            self.be.shard_backward_adam_fused(acc_back, sent, prim, dens, [p[k] for k in SEGMENTS], [view_of(self.exp_avg, k) for k in SEGMENTS],
                                              [view_of(self.exp_avg_sq, k) for k in SEGMENTS], views, self.step_count,
                                              [self.lrs[k] for k in SEGMENTS], self.betas, self.eps)
            return
        grads = tuple(self.grads[k] for k in _BACKWARD_ORDER)
        self.be.shard_backward(acc_back, sent, prim, dens, p['means'], p['scales'], p['rotations'], p['opacities'], p['sh_coefficients_rest'],
                               views, grads)
        segs = [(k, self.layout[k][0], self.layout[k][0] + self.layout[k][1]) for k in SEGMENTS if self.layout[k][1] > 0]
        self.be.adam_step_multi([self.grad_arena[a:b] for _, a, b in segs], [self.param_arena[a:b] for _, a, b in segs],
                                [self.exp_avg[a:b] for _, a, b in segs], [self.exp_avg_sq[a:b] for _, a, b in segs],
                                [self.step_count] * len(segs), [self.lrs[k] for k, _, _ in segs], self.betas[0], self.betas[1], self.eps)

    # ---- public ---------------------------------------------------------------------------------------------------
    def step(self, views: Sequence[RasterizerSettings], target: torch.Tensor, *, update_densification: bool = True) -> torch.Tensor:
        """One optimizer step over the global batch `views` (len == world; view v is rendered by rank v, `target` is the
        ground truth of views[rank]). Every rank must pass the same `views`."""
        assert len(views) == self.world
        G, r = self.world, self.rank
        prim = self.project(views)
        table = self._gather_counts()                                        # [shard, view, (V, I)]
        self.last_counts = table
        sent = [int(table[r, v, 0]) for v in range(G)]                      # my records per view
        got = [int(table[s, r, 0]) for s in range(G)]                       # records of my view per shard
        records = self._all_to_all([self.records[v, :sent[v]] for v in range(G)], got)
        image, acc = self.render(records, int(table[:, r, 1].sum()), views[r], target, shard_counts=got if self.strided else None)
        offs = [0]
        for c in got:
            offs.append(offs[-1] + c)
        acc_back = self._all_to_all([acc[offs[s]:offs[s + 1]] for s in range(G)], sent)
        self.finish(views, prim, acc_back, sent, update_densification)
        return image

    def gather_parameters(self) -> dict:
        """All ranks' shards interleaved back into the global order (export / evaluation; not part of a step)."""
        if self.world == 1:
            return {k: v.clone() for k, v in self.params.items()}
        out = {}
        for k in SEGMENTS:
            mine = self.params[k]
            sizes = torch.tensor([mine.shape[0], int(self.strided)], dtype=torch.int64, device=self.device)
            all_sizes = [torch.empty_like(sizes) for _ in range(self.world)]
            dist.all_gather(all_sizes, sizes, group=self.group)
            rows = [int(s[0]) for s in all_sizes]
            strided = all(int(s[1]) for s in all_sizes)
            pad = max(rows)
            buf = torch.zeros((pad,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=self.device)
            buf[:mine.shape[0]] = mine
            parts = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(parts, buf, group=self.group)
            out[k] = _merge_shards([parts[s][:rows[s]] for s in range(self.world)], strided)
        return out

    def set_learning_rates(self, lrs: dict) -> None:
        self.lrs.update(lrs)

    # ---- maintenance of the shard (densification / pruning / re-ordering run on the owner, every ~100 steps) ---------------
    def as_gaussians(self):
        """The shard as a `harness.trainer.Gaussians` whose optimizer carries this trainer's Adam moments, so that
        `harness.densify` (clone / split / prune / MCMC relocation / opacity reset / Morton sort) can be applied to it unchanged.
        Every decision is per Gaussian and `densification_info` already holds the statistics of ALL views of the steps (the owner
        accumulates them), so owners densify independently -- no collective; only MCMC's global cap needs the total count."""
        from .trainer import Gaussians
        kw = {} if self.max_sh_degree is None else {'max_sh_degree': self.max_sh_degree}
        g = Gaussians({k: self.params[k].clone() for k in SEGMENTS}, self.device, **kw)
        if self.lr_config is not None:
            g.training_setup(training_cameras_extent=self.extent, lr=self.lr_config)
        else:
            g.training_setup(training_cameras_extent=self.extent)
        for group in g.optimizer.param_groups:
            k = group['name']
            o, n, shape = self.layout[k]
            group['lr'] = self.lrs[k]
            g.optimizer.state[group['params'][0]] = {'step': self.step_count, 'exp_avg': self.exp_avg[o:o + n].view(shape).clone(),
                                                     'exp_avg_sq': self.exp_avg_sq[o:o + n].view(shape).clone()}
        g.densification_info = self.densification_info.clone()
        return g

    def rebuild_from(self, g) -> None:
        """Adopt a (densified / pruned / re-ordered) `Gaussians`: new arenas, moments taken from its optimizer state."""
        new = {k: getattr(g, k).detach() for k in SEGMENTS}
        state = {group['name']: g.optimizer.state.get(group['params'][0], {}) for group in g.optimizer.param_groups}
        self.n = new['means'].shape[0]
        self.strided = False
        self.layout, off = {}, 0
        for k in SEGMENTS:
            self.layout[k] = (off, new[k].numel(), tuple(new[k].shape))
            off += (new[k].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        mk = lambda: torch.zeros(off, dtype=torch.float32, device=self.device)
        self.param_arena, self.grad_arena, self.exp_avg, self.exp_avg_sq = mk(), mk(), mk(), mk()
        self.params, self.grads = {}, {}
        for k in SEGMENTS:
            o, n, shape = self.layout[k]
            self.params[k] = self.param_arena[o:o + n].view(shape)
            self.params[k].copy_(new[k])
            self.grads[k] = self.grad_arena[o:o + n].view(shape)
            if 'exp_avg' in state[k]:
                self.exp_avg[o:o + n].view(shape).copy_(state[k]['exp_avg'])
                self.exp_avg_sq[o:o + n].view(shape).copy_(state[k]['exp_avg_sq'])
        self.densification_info = g.densification_info.clone() if g.densification_info is not None and g.densification_info.shape[1] == self.n \
            else torch.zeros((2, self.n), dtype=torch.float32, device=self.device)
        self.records = torch.empty((self.records.shape[0], max(self.n, 1), _lib.SPLAT_RECORD_BYTES), dtype=torch.uint8, device=self.device)


class LocalShardGroup:
    """All G shard owners of a step inside ONE process, exchanging through local copies instead of RCCL: the functional twin of
    G ranks. Used by the tests (multi-shard logic on a single GPU) and by tools/sharded_emulation.py, which times it to get the
    per-rank COMPUTE cost of the sharded step (total / G) next to the single-GPU iteration."""

    def __init__(self, backend: Backend, params: dict, lrs: dict, world: int, **kw) -> None:
        self.world = world
        self.ranks = [ShardedTrainer(backend, shard_of(params, r, world), lrs, **kw) for r in range(world)]
        for r, t in enumerate(self.ranks):
            t.world, t.rank = world, r                      # loss scaling 1/G and record buffers for G views
            t.records = torch.empty((world, max(t.n, 1), _lib.SPLAT_RECORD_BYTES), dtype=torch.uint8, device=t.device)
            t.counts = torch.zeros((world, 2), dtype=torch.int32, device=t.device)

    def step(self, views: Sequence[RasterizerSettings], targets: Sequence[torch.Tensor], *, update_densification: bool = True) -> list:
        G = self.world
        prims = [t.project(views) for t in self.ranks]
        table = torch.stack([t.counts for t in self.ranks]).cpu()            # [shard, view, (V, I)]
        images, accs = [], []
        for v in range(G):
            records = torch.cat([self.ranks[s].records[v, :int(table[s, v, 0])] for s in range(G)], dim=0)
            image, acc = self.ranks[v].render(records, int(table[:, v, 1].sum()), views[v], targets[v],
                                              shard_counts=[int(table[s, v, 0]) for s in range(G)] if all(t.strided for t in self.ranks) else None)
            images.append(image)
            accs.append(acc)
        for s in range(G):
            sent = [int(table[s, v, 0]) for v in range(G)]
            pieces = []
            for v in range(G):
                o = int(table[:s, v, 0].sum())
                pieces.append(accs[v][o:o + sent[v]])
            self.ranks[s].finish(views, prims[s], torch.cat(pieces, dim=0), sent, update_densification)
        self.last_counts = table
        return images

    def gather_parameters(self) -> dict:
        out = {}
        for k in SEGMENTS:
            parts = [t.params[k] for t in self.ranks]
            out[k] = _merge_shards(parts, all(t.strided for t in self.ranks))
        return out
