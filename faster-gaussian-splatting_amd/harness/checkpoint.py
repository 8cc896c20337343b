"""Training checkpoints: everything a resumed run needs to continue bit-identically -- the six parameter tensors, the Adam
moments and step counts, the active SH degree, the densification statistics and the scene extent / learning-rate table.
(The reference inherits checkpointing from NeRFICG's BaseModel / BaseTrainer, which are not vendored: `torch.save` of the
module and optimizer state dicts; this is the same content for the harness's `Gaussians`.)"""
from __future__ import annotations

import torch

from .trainer import PARAM_ORDER, Gaussians


@torch.no_grad()
def save_checkpoint(g: Gaussians, path, iteration: int = 0) -> None:
    state = {'format': 'fgs-checkpoint-1', 'iteration': int(iteration), 'max_sh_degree': g.max_sh_degree,
             'active_sh_degree': g.active_sh_degree, 'params': {k: getattr(g, k).detach().cpu() for k in PARAM_ORDER},
             'densification_info': None if g.densification_info is None else g.densification_info.cpu(),
             'extent': getattr(g, 'extent', None), 'lr': getattr(g, '_lr', None), 'optimizer': None}
    if getattr(g, 'optimizer', None) is not None:
        opt = {}
        for group in g.optimizer.param_groups:
            st = g.optimizer.state.get(group['params'][0], {})
            opt[group['name']] = {'lr': group['lr'], 'step': st.get('step', 0),
                                  'exp_avg': None if 'exp_avg' not in st else st['exp_avg'].cpu(),
                                  'exp_avg_sq': None if 'exp_avg_sq' not in st else st['exp_avg_sq'].cpu()}
        state['optimizer'] = opt
    torch.save(state, path)


def load_checkpoint(path, device) -> tuple[Gaussians, int]:
    state = torch.load(path, map_location='cpu', weights_only=True)      # tensors + plain containers only: never unpickle arbitrary objects
    if state.get('format') != 'fgs-checkpoint-1':
        raise ValueError(f'{path} is not an fgs checkpoint')
    g = Gaussians(state['params'], device, max_sh_degree=state['max_sh_degree'], active_sh_degree=state['active_sh_degree'])
    # None = the state after the last densification iteration (Trainer.py:120-145): the constructor's zeros must not come back
    g.densification_info = None if state['densification_info'] is None else state['densification_info'].to(device)
    if state['optimizer'] is not None:
        g.training_setup(training_cameras_extent=state['extent'], lr=state['lr'])
        for group in g.optimizer.param_groups:
            saved = state['optimizer'][group['name']]
            group['lr'] = saved['lr']
            if saved['exp_avg'] is not None:
                g.optimizer.state[group['params'][0]] = {'step': saved['step'], 'exp_avg': saved['exp_avg'].to(device).contiguous(),
                                                         'exp_avg_sq': saved['exp_avg_sq'].to(device).contiguous()}
    return g, state['iteration']
