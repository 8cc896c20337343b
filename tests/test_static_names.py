"""The driver-facing scripts (bench.py, __graft_entry__.py) and the GPU-only tools cannot be executed in the build container, and large parts
of bench.py run only with N > 1 GPUs: an undefined name there is found by the round-end run, as rc != 0. This is a stdlib-only
undefined-name check (names loaded in a function that are neither bound in it or an enclosing function, nor module globals, nor builtins)."""
import ast
import builtins
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
FILES = ['bench.py', '__graft_entry__.py'] + sorted(str(p.relative_to(ROOT)) for p in (ROOT / 'tools').glob('*.py'))


def _bound_names(node) -> set:
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            out |= {a.asname or a.name.split('.')[0] for a in n.names}
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            args = n.args
            out |= {a.arg for a in args.args + args.kwonlyargs + args.posonlyargs}
            out |= {a.arg for a in (args.vararg, args.kwarg) if a is not None}
    return out


@pytest.mark.parametrize('rel', FILES)
def test_no_undefined_names(rel):
    tree = ast.parse((ROOT / rel).read_text())
    known = _bound_names(tree) | set(dir(builtins)) | {'__file__', '__name__', '__doc__'}      # every name bound anywhere in the file
    loaded = {(n.id, n.lineno) for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    missing = sorted((name, line) for name, line in loaded if name not in known)
    assert not missing, f'{rel}: names that are never bound: {missing}'
