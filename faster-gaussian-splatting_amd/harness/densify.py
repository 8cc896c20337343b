"""Caller-side maintenance of the Gaussian set, following the reference's Model.py: adaptive density control
(clone / split / prune), pruning, re-ordering, Morton order, opacity reset, SH degree schedule -- including the Adam-state surgery that
NeRFICG's `Optim.adam_utils` helpers perform in the reference (`extend_param_groups`, `prune_param_groups`, `sort_param_groups`,
`replace_param_group_data`; not vendored, semantics as in the original 3DGS code base: new entries start with zero moments,
pruned/sorted entries keep theirs, replaced data resets its moments).

Adaptive density control runs as the device passes of csrc/densify.hip (classify -> scan -> one scatter of the 59 parameters and 2 x 59
moments per Gaussian): every tensor is read once and written once; there is no second, torch-op formulation of it in the product (the
readable numpy restatement of Model.py:312-366 that the tests compare with lives with the test infrastructure). Prune / sort go through one gather
launch for all 18 tensors, the Morton order through key + radix sort; for CPU tensors (host-side tooling, the CPU tests) prune / sort fall back
to plain indexing. A backend other than the HIP library -- the CPU simulation in the tests -- is injected with `ops_backend`.

Runs every 100 iterations (Trainer.py:120-139); consumes the `densification_info[2,N]` statistics that the backward pass accumulates
(kernels_backward.cuh:194-201).
"""
from __future__ import annotations

import math

import torch

from .scenes import morton_order
from .trainer import PARAM_ORDER, Gaussians

GARDEN_SCHEDULE = {  # fastergs_garden.yaml:66-70 / Trainer.py:16-67
    'densification_start': 600, 'densification_end': 14_900, 'densification_interval': 100, 'grad_threshold': 2.0e-4,
    'percent_dense': 0.01, 'opacity_reset_interval': 3_000, 'morton_interval': 5_000, 'morton_end': 15_000, 'sh_interval': 1_000,
}


def _device_backend(g: Gaussians, ops_backend=None):
    """The backend whose densify.hip passes serve `g`: the one given, else the HIP library for device tensors, else None (torch ops)."""
    if ops_backend is not None:
        return ops_backend
    if g.means.is_cuda:
        from FasterGSCudaBackend._backend import default_backend
        return default_backend()
    return None


def _complete(st) -> bool:
    return bool(st) and st.get('exp_avg') is not None and st.get('exp_avg_sq') is not None


def ensure_state(g: Gaussians) -> None:
    """Mixed optimizer state (a checkpoint saved a group without moments, or a group never received a gradient): the groups WITH state must keep
    their moments row-aligned with the parameters through a gather / scatter, so the others get the state Adam would create lazily -- zero moments,
    step 0 -- instead of everything being rebound without moments (stale row counts -> out-of-bounds reads in the next step). Explicit (round-3
    advisor finding: this used to happen inside the getter below); a no-op when no group or every group has state."""
    opt = g.optimizer
    if opt is None:
        return
    states = [opt.state.get(group['params'][0]) for group in opt.param_groups]
    if not any(_complete(st) for st in states):
        return
    for group, st in zip(opt.param_groups, states):
        if not _complete(st):
            param = group['params'][0]
            opt.state[param] = {'step': (st or {}).get('step', 0), 'exp_avg': torch.zeros_like(param), 'exp_avg_sq': torch.zeros_like(param)}


def _moments(g: Gaussians):
    """(exp_avgs, exp_avg_sqs) in PARAM_ORDER if every group has optimizer state, else (None, None). Pure: call ensure_state() first where a mixed
    state must be completed."""
    opt = g.optimizer
    if opt is None:
        return None, None
    by_name = {group['name']: opt.state.get(group['params'][0]) for group in opt.param_groups}
    if not all(_complete(by_name.get(k)) for k in PARAM_ORDER):
        return None, None
    return [by_name[k]['exp_avg'] for k in PARAM_ORDER], [by_name[k]['exp_avg_sq'] for k in PARAM_ORDER]


def _adopt(g: Gaussians, new_params, new_m, new_v) -> None:
    """Installs new parameter tensors (PARAM_ORDER lists) and, if given, their moments; step counts are kept."""
    pm = dict(zip(PARAM_ORDER, new_params))
    mm = dict(zip(PARAM_ORDER, new_m)) if new_m is not None else None
    vm = dict(zip(PARAM_ORDER, new_v)) if new_v is not None else None
    if mm is not None:
        _rebind(g, pm, None, moments=(mm, vm))
    else:
        _rebind(g, pm, lambda s, k: s)


def _rebind(g: Gaussians, new: dict, state_fn, moments=None) -> None:
    """Replaces the six parameters (and their optimizer state via state_fn(old_state_tensor, name)) in place."""
    opt = g.optimizer
    for group in (opt.param_groups if opt is not None else []):
        name = group['name']
        old = group['params'][0]
        param = torch.nn.Parameter(new[name].contiguous())
        state = opt.state.pop(old, None)
        if state and moments is not None:
            opt.state[param] = {'step': state['step'], 'exp_avg': moments[0][name], 'exp_avg_sq': moments[1][name]}
        elif state:
            opt.state[param] = {'step': state['step'], 'exp_avg': state_fn(state['exp_avg'], name).contiguous(),
                                'exp_avg_sq': state_fn(state['exp_avg_sq'], name).contiguous()}
        group['params'][0] = param
        setattr(g, name, param)
    if opt is None:
        for name in PARAM_ORDER:
            setattr(g, name, torch.nn.Parameter(new[name].contiguous()))


def extend(g: Gaussians, extra: dict) -> None:
    """extend_param_groups: append new Gaussians, their Adam moments start at zero."""
    new = {k: torch.cat([getattr(g, k).detach(), extra[k]]) for k in PARAM_ORDER}
    _rebind(g, new, lambda s, k: torch.cat([s, torch.zeros_like(extra[k])]))


def _gather(g: Gaussians, index: torch.Tensor, be) -> None:
    """Parameters and both moments through an index list in ONE gather launch (csrc/densify.hip)."""
    params = [getattr(g, k).detach() for k in PARAM_ORDER]
    ensure_state(g)
    m, v = _moments(g)
    outs = be.gather_rows(params + (m or []) + (v or []), index)
    _adopt(g, outs[:6], outs[6:12] if m else None, outs[12:18] if m else None)
    if g.densification_info is not None:
        g.densification_info = g.densification_info[:, index].contiguous()


def prune(g: Gaussians, prune_mask: torch.Tensor, ops_backend=None) -> None:
    """Model.py:275-291."""
    keep = ~prune_mask
    be = _device_backend(g, ops_backend)
    if be is not None:
        return _gather(g, torch.nonzero(keep).flatten(), be)
    _rebind(g, {k: getattr(g, k).detach()[keep] for k in PARAM_ORDER}, lambda s, k: s[keep])
    if g.densification_info is not None:
        g.densification_info = g.densification_info[:, keep].contiguous()


def sort(g: Gaussians, ordering: torch.Tensor, ops_backend=None) -> None:
    """Model.py:293-306."""
    be = _device_backend(g, ops_backend)
    if be is not None:
        return _gather(g, ordering, be)
    _rebind(g, {k: getattr(g, k).detach()[ordering] for k in PARAM_ORDER}, lambda s, k: s[ordering])
    if g.densification_info is not None:
        g.densification_info = g.densification_info[:, ordering].contiguous()


def apply_morton_ordering(g: Gaussians, ops_backend=None) -> None:
    """Model.py:459-463 (CudaUtils.MortonEncoding is not vendored: the 10-bit-per-axis Z-curve of harness.scenes.morton_order, as
    key + radix-sort passes on the device)."""
    be = _device_backend(g, ops_backend)
    if be is not None:
        return sort(g, be.morton_order(g.means.detach()), be)
    sort(g, morton_order(g.means.detach().cpu()).to(g.means.device))


def reset_densification_info(g: Gaussians) -> None:
    g.densification_info = torch.zeros((2, g.means.shape[0]), dtype=torch.float32, device=g.means.device)   # Model.py:308-310


def reset_opacities(g: Gaussians) -> None:
    """Model.py:262-273 (3D filter off): clamp to sigmoid^-1(0.01) and reset the opacity group's moments."""
    new_op = g.opacities.detach().clamp_max(-4.595119953155518)
    new = {k: getattr(g, k).detach() for k in PARAM_ORDER}
    new['opacities'] = new_op
    _rebind(g, new, lambda s, k: torch.zeros_like(s) if k == 'opacities' else s)


def adaptive_density_control(g: Gaussians, grad_threshold: float, min_opacity: float, prune_large_gaussians: bool,
                             percent_dense: float = 0.01, generator: torch.Generator | None = None, ops_backend=None) -> dict:
    """Model.py:312-366: clone small / split large Gaussians whose mean screen-space gradient exceeds the threshold, then prune."""
    be = _device_backend(g, ops_backend)
    if be is not None:
        params = [getattr(g, k).detach() for k in PARAM_ORDER]
        ensure_state(g)
        m, v = _moments(g)
        dev = g.means.device
        noise_fn = None
        if generator is not None:      # reproducible runs: the samples come from the given generator (CPU generators feed the device tensor)
            noise_fn = lambda rows: torch.randn((rows, 3), generator=generator, device=generator.device).to(dev)
        new_p, new_m, new_v, (kept, clones, children, split) = be.adaptive_density_control(
            g.densification_info, params, m, v, grad_threshold, min_opacity, prune_large_gaussians, percent_dense, g.extent, noise_fn)
        n_old = params[0].shape[0]
        _adopt(g, new_p, new_m, new_v)
        g.densification_info = None                                                        # Model.py:353-355
        # cloned / children_per_copy count what SURVIVED the pruning that follows the densification; pruned = old Gaussians that are gone
        return {'cloned': clones, 'split': split, 'pruned': n_old - kept, 'total': g.means.shape[0], 'kept': kept, 'children_per_copy': children}
    raise RuntimeError('adaptive_density_control: the Gaussians live on the CPU and no backend was given -- density control exists as the device '
                       'passes of csrc/densify.hip only (pass ops_backend=..., or move the model to the ROCm device)')


def run_callbacks(g: Gaussians, iteration: int, schedule: dict = GARDEN_SCHEDULE, generator: torch.Generator | None = None, ops_backend=None) -> dict | None:
    """The per-iteration schedule of Trainer.py:114-165 in priority order (SH degree 110, densify 100, Morton 99, opacity
    reset 90); call BEFORE training_iteration(iteration) like the reference's callback dispatcher does."""
    s, out = schedule, None
    if iteration >= s['sh_interval'] and iteration % s['sh_interval'] == 0:
        g.increase_used_sh_degree()
    if s['densification_start'] <= iteration <= s['densification_end'] and (iteration - s['densification_start']) % s['densification_interval'] == 0:
        out = adaptive_density_control(g, s['grad_threshold'], 0.005, iteration > s['opacity_reset_interval'], s['percent_dense'], generator, ops_backend)
        if iteration < s['densification_end']:
            reset_densification_info(g)
    if iteration <= s['morton_end'] and iteration % s['morton_interval'] == 0:
        apply_morton_ordering(g, ops_backend)
    if s['opacity_reset_interval'] <= iteration <= s['densification_end'] and iteration % s['opacity_reset_interval'] == 0:
        reset_opacities(g)
    return out


# ---- the MCMC policy and Speedy-Splat pruning (reference Model.py:367-457,465-484); their kernels are the backend's
# relocation_adjustment / add_noise / update_pruning_scores. `ops` injects the operators (default: the HIP backend's wrappers;
# the CPU tests pass the simulation backend's).
def _default_ops():
    from FasterGSCudaBackend import add_noise, relocation_adjustment
    return relocation_adjustment, add_noise


def reset_state(g: Gaussians, indices: torch.Tensor) -> None:
    """Optim.adam_utils.reset_state: zero the Adam moments of the given Gaussians in every group."""
    if g.optimizer is None:
        return
    for group in g.optimizer.param_groups:
        st = g.optimizer.state.get(group['params'][0])
        if st:
            st['exp_avg'][indices] = 0.0
            st['exp_avg_sq'][indices] = 0.0


def _dead_mask(g: Gaussians, min_opacity: float) -> torch.Tensor:
    """Gaussians the MCMC policy recycles: opacity at or below the threshold, or a quaternion that no longer encodes a rotation."""
    logit_floor = math.log(min_opacity / (1.0 - min_opacity))
    return (g.opacities.detach().flatten() <= logit_floor) | (g.rotations.detach().square().sum(dim=1) < 1e-8)


def _draw_by_opacity(g: Gaussians, candidates: torch.Tensor | None, how_many: int, generator) -> torch.Tensor:
    """`how_many` indices drawn with replacement, with probability proportional to the activated opacity (over `candidates` or over everything).
    The draw happens on the host so that a seeded CPU generator reproduces a run on any device."""
    weight = torch.sigmoid(g.opacities.detach()).flatten()
    if candidates is not None:
        weight = weight[candidates]
    picks = torch.multinomial(weight.cpu(), how_many, replacement=True, generator=generator).to(g.means.device)
    return picks if candidates is None else candidates[picks]


@torch.no_grad()
def _split_mass(g: Gaussians, sources: torch.Tensor, min_opacity: float, relocation_adjustment):
    """3DGS-MCMC Eq. 9 through the backend's kernel: a source drawn c times ends up as c + 1 Gaussians at the same place, and each of them gets
    the opacity / scale that keeps the rendered result (Model.py:382-391 / 424-433 semantics). Returns (opacity logits [k, 1], log scales [k, 3])
    for the `sources` list, the same value for every occurrence of a source."""
    multiplicity = torch.bincount(sources, minlength=g.means.shape[0])[sources] + 1
    activated = torch.sigmoid(g.opacities.detach()).flatten()[sources].reshape(-1, 1).contiguous()
    extents = g.scales.detach()[sources].exp().contiguous()
    shared_opacity, shared_extent = relocation_adjustment(activated, extents, multiplicity)
    ceiling = 1.0 - torch.finfo(torch.float32).eps
    return shared_opacity.clamp(min_opacity, ceiling).logit().reshape(-1, 1), shared_extent.log()


def _forget_moments_and_statistics(g: Gaussians, rows: torch.Tensor) -> None:
    reset_state(g, rows)
    g.densification_info = None


@torch.no_grad()
def mcmc_densification(g: Gaussians, min_opacity: float, cap_max: int, generator: torch.Generator | None = None, ops=None) -> dict:
    """The MCMC policy's densification step (Model.py:367-457 semantics): (1) every dead Gaussian is moved onto a live one drawn by opacity, source
    and copies sharing opacity and scale; (2) the set then grows by 5 % (up to `cap_max`) with copies of Gaussians drawn the same way. Sources lose
    their Adam moments, copies start without any."""
    relocation_adjustment, _ = ops or _default_ops()
    report = {'relocated': 0, 'added': 0}
    untouched = ('means', 'sh_coefficients_0', 'sh_coefficients_rest', 'rotations')          # what a copy inherits as it is
    dead = _dead_mask(g, min_opacity)
    graves = dead.nonzero().flatten()
    if graves.numel() > 0:
        sources = _draw_by_opacity(g, (~dead).nonzero().flatten(), graves.numel(), generator)
        logit, log_scale = _split_mass(g, sources, min_opacity, relocation_adjustment)
        for rows in (sources, graves):
            g.opacities.data[rows] = logit
            g.scales.data[rows] = log_scale
        for name in untouched:
            tensor = getattr(g, name).data
            tensor[graves] = tensor[sources]
        _forget_moments_and_statistics(g, sources)
        report['relocated'] = int(graves.numel())
    current = g.means.shape[0]
    growth = min(cap_max, int(1.05 * current)) - current
    if growth > 0:
        sources = _draw_by_opacity(g, None, growth, generator)
        logit, log_scale = _split_mass(g, sources, min_opacity, relocation_adjustment)
        g.opacities.data[sources] = logit
        g.scales.data[sources] = log_scale
        newcomers = {name: getattr(g, name).detach()[sources] for name in untouched}
        newcomers.update(opacities=logit, scales=log_scale)
        extend(g, newcomers)
        _forget_moments_and_statistics(g, sources)
        report['added'] = growth
    report['total'] = g.means.shape[0]
    return report


@torch.no_grad()
def importance_pruning(g: Gaussians, scores: torch.Tensor, pruning_ratio: float) -> int:
    """Model.py:465-470 (Speedy-Splat): drop the given fraction of Gaussians with the lowest accumulated importance score."""
    k = int(pruning_ratio * (scores.numel() - 1)) + 1                          # kthvalue is 1-based
    threshold = torch.kthvalue(scores.cpu(), k).values.to(scores.device)
    mask = scores <= threshold
    prune(g, mask)
    return int(mask.sum())


@torch.no_grad()
def post_optimizer_step(g: Gaussians, inject_noise: bool, lr_means: float, ops=None) -> None:
    """Model.py:472-477 (3D filter off): the SGLD noise of 3DGS-MCMC after every optimizer step."""
    if inject_noise:
        _, add_noise = ops or _default_ops()
        add_noise(g.scales.detach(), g.rotations.detach(), g.opacities.detach(), g.means.data, 5e5 * lr_means)


# ---- a whole (possibly time-compressed) training run from a random initialisation: Trainer.py:86-201 end to end ---------------------------------
def run_mcmc_callbacks(g: Gaussians, iteration: int, schedule: dict, cap_max: int, generator: torch.Generator | None = None, ops=None, ops_backend=None) -> dict | None:
    """The callbacks of Trainer.py:114-165 under USE_MCMC: SH degree, `mcmc_densification` in the densification window, Morton order; no
    opacity reset (Trainer.py:154-165 skip it) and no densification statistics."""
    s, out = schedule, None
    if iteration >= s['sh_interval'] and iteration % s['sh_interval'] == 0:
        g.increase_used_sh_degree()
    if s['densification_start'] <= iteration <= s['densification_end'] and (iteration - s['densification_start']) % s['densification_interval'] == 0:
        out = mcmc_densification(g, 0.005, cap_max, generator, ops)
    if iteration <= s['morton_end'] and iteration % s['morton_interval'] == 0:
        apply_morton_ordering(g, ops_backend)
    return out


def add_mcmc_regularisation_gradients(g: Gaussians, lambda_opacity: float, lambda_scale: float) -> None:
    """d/d(parameters) of  lambda_opacity * mean(sigmoid(opacities)) + lambda_scale * mean(exp(scales))  (Loss.py:17-18, Model.py:136-142; 0.01 each
    with MCMC, fastergs_garden.yaml:100-101), added to the photometric gradients between backward and optimizer step."""
    with torch.no_grad():
        if lambda_opacity:
            a = torch.sigmoid(g.opacities.detach())
            g.opacities.grad.add_(a * (1.0 - a), alpha=lambda_opacity / a.numel())
        if lambda_scale:
            e = torch.exp(g.scales.detach())
            g.scales.grad.add_(e, alpha=lambda_scale / e.numel())


def train_from_scratch(views, targets, bbox_lo: torch.Tensor, bbox_hi: torch.Tensor, *, n_points: int = 100_000, iterations: int = 30_000,
                       schedule_scale: float = 1.0, seed: int = 7, max_gaussians: int = 0, on_iteration=None, policy: str = 'adc',
                       max_primitives: int = 1_000_000, ops=None) -> tuple[Gaussians, dict]:
    """RANDOM_INITIALIZATION of fastergs_garden.yaml (N_POINTS uniform samples of the bounding box, carved to the points inside at least one training
    frustum: utils.py:29-52; Model.py:202-231) and the garden schedule with every interval multiplied by `schedule_scale` (1.0 = the reference's
    30 000 iterations; bench.py's `trained_like` block runs a tenth), random view order (Trainer.py:84), through `run_callbacks` on the tensors'
    device. `policy='mcmc'` is the configuration's USE_MCMC variant (the comments of fastergs_garden.yaml:69,79-83,100-101): MCMC initialisation
    (scales x 0.1, opacity 0.5), relocation + 5 % growth per densification step up to `max_primitives`, densification until 24 900 and Morton order
    until 25 000, no opacity reset, SGLD noise after every optimizer step, opacity and scale regularisation 0.01.
    Returns the trained Gaussians and {'count_curve': [[iteration, count], ...], 'extent': ...}."""
    assert policy in ('adc', 'mcmc')
    mcmc = policy == 'mcmc'
    from .scenes import initialize_from_point_cloud
    from .trainer import training_iteration
    dev = views[0].w2c.device
    gen = torch.Generator().manual_seed(seed)
    pts = (torch.rand((n_points, 3), generator=gen) * (bbox_hi - bbox_lo).cpu() + bbox_lo.cpu()).to(dev)
    seen = torch.zeros(n_points, dtype=torch.bool, device=dev)
    for v in views:
        cam = pts @ v.w2c[:3, :3].T + v.w2c[:3, 3]
        z = cam[:, 2]
        x, y = cam[:, 0] / z * v.focal_x + v.center_x, cam[:, 1] / z * v.focal_y + v.center_y
        seen |= (z > v.near_plane) & (z < v.far_plane) & (x >= 0) & (x < v.width) & (y >= 0) & (y < v.height)
    g = Gaussians(initialize_from_point_cloud(pts[seen].contiguous(), use_mcmc=mcmc), dev, active_sh_degree=0)
    centers = torch.stack([v.position for v in views])
    extent = float(1.1 * (centers - centers.mean(dim=0)).norm(dim=1).max())                 # Trainer.py:91
    lr = dict(__import__('harness.trainer', fromlist=['GARDEN_LR']).GARDEN_LR)
    lr['means_max_steps'] = max(1, int(round(lr['means_max_steps'] * schedule_scale)))
    g.training_setup(training_cameras_extent=extent, lr=lr)
    schedule = dict(GARDEN_SCHEDULE)
    if mcmc:
        schedule.update(densification_end=24_900, morton_end=25_000)
        ensure_state(g)
        regularise = lambda: add_mcmc_regularisation_gradients(g, 0.01, 0.01)
    else:
        reset_densification_info(g)
    for k in ('densification_start', 'densification_end', 'densification_interval', 'opacity_reset_interval', 'morton_interval', 'morton_end', 'sh_interval'):
        schedule[k] = max(1, int(round(schedule[k] * schedule_scale)))
    dgen = torch.Generator().manual_seed(seed + 1)
    curve, order = [[0, g.means.shape[0]]], []
    for it in range(iterations):
        if max_gaussians and g.means.shape[0] >= max_gaussians:
            schedule['grad_threshold'] = float('inf')               # a budget guard (not in the reference): keeps pruning, stops cloning / splitting
        stats = run_mcmc_callbacks(g, it, schedule, max_primitives, dgen, ops) if mcmc else run_callbacks(g, it, schedule, dgen)
        if stats:
            curve.append([it, stats['total']])
        if it % len(views) == 0:
            order = torch.randperm(len(views), generator=gen).tolist()
        v = order[it % len(views)]
        if mcmc:
            loss = training_iteration(g, views[v], targets[v], it, densification_end=0, before_step=regularise)
            post_optimizer_step(g, True, next(pg['lr'] for pg in g.optimizer.param_groups if pg['name'] == 'means'), ops)
        else:
            loss = training_iteration(g, views[v], targets[v], it, densification_end=schedule['densification_end'])
        if on_iteration is not None:
            on_iteration(it, loss)
    return g, {'count_curve': curve, 'extent': extent, 'schedule': schedule}
