"""On-disk format: the 3DGS-compatible PLY layout of the reference's `Gaussians.as_ply_dict`
(Model.py:511-542): x,y,z, f_dc_0..2, f_rest_* (channel-major), opacity (logit), scale_* (log), rot_* (normalised, w first),
all float32, binary little endian. `load_ply` is the inverse (the reference delegates reading to NeRFICG)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from .trainer import PARAM_ORDER, Gaussians


def attribute_names(n_rest: int) -> list[str]:
    return (['x', 'y', 'z'] + ['f_dc_0', 'f_dc_1', 'f_dc_2'] + [f'f_rest_{i}' for i in range(3 * n_rest)] + ['opacity']
            + ['scale_0', 'scale_1', 'scale_2'] + ['rot_0', 'rot_1', 'rot_2', 'rot_3'])


@torch.no_grad()
def as_ply_dict(g: Gaussians) -> dict:
    """Model.py:511-542."""
    if g.means.shape[0] == 0:
        return {}
    rot = g.rotations.detach()
    rot = rot / rot.norm(dim=1, keepdim=True)                           # `self.rotations` is the normalised quaternion
    cols = [g.means.detach(), g.sh_coefficients_0.detach().transpose(1, 2).flatten(start_dim=1),
            g.sh_coefficients_rest.detach().transpose(1, 2).flatten(start_dim=1), g.opacities.detach(), g.scales.detach(), rot]
    attributes = np.concatenate([c.contiguous().cpu().numpy() for c in cols], axis=1).astype('<f4')
    names = attribute_names(g.sh_coefficients_rest.shape[1])
    vertices = np.empty(attributes.shape[0], dtype=[(n, '<f4') for n in names])
    for i, n in enumerate(names):
        vertices[n] = attributes[:, i]
    return {'vertex': vertices, 'comments': ['SplatRenderMode: default', 'Generated with faster-gaussian-splatting_amd']}   # Model.py:574-576


def save_ply(g: Gaussians, path) -> None:
    data = as_ply_dict(g)
    v = data.get('vertex', np.empty(0, dtype=[(n, '<f4') for n in attribute_names(g.sh_coefficients_rest.shape[1])]))
    header = ['ply', 'format binary_little_endian 1.0'] + [f'comment {c}' for c in data.get('comments', [])]
    header += [f'element vertex {v.shape[0]}'] + [f'property float {n}' for n in v.dtype.names] + ['end_header']
    with open(path, 'wb') as f:
        f.write(('\n'.join(header) + '\n').encode('ascii'))
        f.write(v.tobytes())


def load_ply(path, device='cpu') -> dict:
    raw = Path(path).read_bytes()
    end = raw.index(b'end_header\n') + len(b'end_header\n')
    lines = raw[:end].decode('ascii').splitlines()
    assert lines[0] == 'ply' and lines[1] == 'format binary_little_endian 1.0'
    n = int(next(l for l in lines if l.startswith('element vertex')).split()[-1])
    names = [l.split()[-1] for l in lines if l.startswith('property float')]
    v = np.frombuffer(raw, dtype=[(k, '<f4') for k in names], count=n, offset=end)
    n_rest = sum(k.startswith('f_rest_') for k in names) // 3
    col = lambda ks: torch.from_numpy(np.stack([v[k] for k in ks], axis=1).astype(np.float32))
    out = {
        'means': col(['x', 'y', 'z']),
        'sh_coefficients_0': col(['f_dc_0', 'f_dc_1', 'f_dc_2']).reshape(n, 3, 1).transpose(1, 2),
        'sh_coefficients_rest': col([f'f_rest_{i}' for i in range(3 * n_rest)]).reshape(n, 3, n_rest).transpose(1, 2),
        'opacities': col(['opacity']), 'scales': col(['scale_0', 'scale_1', 'scale_2']), 'rotations': col(['rot_0', 'rot_1', 'rot_2', 'rot_3']),
    }
    return {k: out[k].contiguous().to(device) for k in PARAM_ORDER}
