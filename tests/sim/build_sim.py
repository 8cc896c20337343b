"""Builds tests/sim/_build/libfgs_sim.so: the product's .hip sources compiled by g++ against the stand-in headers
(see README.md). Test infrastructure only."""
from __future__ import annotations

import subprocess
from pathlib import Path

SIM = Path(__file__).resolve().parent
REPO = SIM.parent.parent
CSRC = REPO / 'faster-gaussian-splatting_amd' / 'csrc'
OUT = SIM / '_build' / 'libfgs_sim.so'


def build(force: bool = False) -> Path:
    srcs = sorted(CSRC.glob('*.hip'))
    deps = srcs + sorted(CSRC.glob('*.h')) + sorted(SIM.glob('include/**/*.h*')) + [REPO / 'include' / 'fgs_hip.h']
    if not force and OUT.exists() and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return OUT
    OUT.parent.mkdir(exist_ok=True)
    cmd = ['g++', '-std=c++17', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
           '-Wno-unknown-pragmas', '-Wno-sign-compare', '-Wno-unused-variable', '-Wno-unused-but-set-variable',
           f'-I{SIM / "include"}', f'-I{CSRC}', f'-I{REPO / "include"}', '-o', str(OUT)]
    for s in srcs:
        cmd += ['-x', 'c++', str(s)]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
