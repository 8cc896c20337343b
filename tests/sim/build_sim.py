"""Builds tests/sim/_build/libfgs_sim.so: the product's .hip sources compiled by g++ against the stand-in headers
(see README.md). Test infrastructure only. The translation units are compiled in parallel (one g++ per source) and linked."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

SIM = Path(__file__).resolve().parent
REPO = SIM.parent.parent
CSRC = REPO / 'faster-gaussian-splatting_amd' / 'csrc'
OUT = SIM / '_build' / 'libfgs_sim.so'
PRODUCT_OUT = SIM / '_build' / 'product' / 'libfgs_sim_product.so'      # the same sources WITHOUT -DFGS_DEV_SWITCHES: the product library's host paths
FLAGS = ['-std=c++17', '-O2', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function', '-Wno-unknown-pragmas', '-Wno-sign-compare',
         '-Wno-unused-variable', '-Wno-unused-but-set-variable', '-Wno-attributes', f'-I{SIM / "include"}', f'-I{CSRC}', f'-I{REPO / "include"}']
# FGS_SIM_SANITIZE=undefined: an occasional deep check -- the kernels' integer / shift / alignment / bounds-of-static-array behaviour under UBSan
# (reports go to stderr, the run continues); rebuild with `python tests/sim/build_sim.py` afterwards to get the plain library back
# FGS_SIM_DEFINES="-DFGS_PREPROCESS_ITEMS=2 ...": compile-time experiments of the product sources, checked on the simulation before a GPU A/B
DEV_FLAGS = ['-DFGS_DEV_SWITCHES']   # the simulation carries the dev build's A/B variants and switches (the sim tests compare formulations);
                                     # build(product=True) leaves them out: the switches are constants there, as in libfgs_hip.so
FLAGS += os.environ.get('FGS_SIM_DEFINES', '').split()
SAN = [f'-fsanitize={os.environ["FGS_SIM_SANITIZE"]}', '-fsanitize-recover=all', '-g'] if os.environ.get('FGS_SIM_SANITIZE') else []


def build(force: bool = False, product: bool = False) -> Path:
    """(Re)builds the library if a source or header is newer; safe under pytest-xdist: concurrent callers serialise on a lock file, so no worker
    ever dlopens a half-linked library. product=True: the flavour without the dev switches (its own directory and file)."""
    import fcntl
    out = PRODUCT_OUT if product else OUT
    out.parent.mkdir(parents=True, exist_ok=True)
    with open(out.parent / '.build.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, out, [] if product else DEV_FLAGS)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, OUT: Path, flavour: list) -> Path:
    srcs = sorted(CSRC.glob('*.hip'))
    headers = sorted(CSRC.glob('*.h')) + sorted(SIM.glob('include/**/*.h*')) + [REPO / 'include' / 'fgs_hip.h']
    newest_header = max(h.stat().st_mtime for h in headers)

    def compile_one(src: Path) -> Path:
        obj = OUT.parent / (src.stem + '.o')
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, newest_header):
            subprocess.run(['g++', *FLAGS, *flavour, *SAN, '-c', '-x', 'c++', str(src), '-o', str(obj)], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, srcs))
    if force or not OUT.exists() or any(OUT.stat().st_mtime < o.stat().st_mtime for o in objs):
        tmp = OUT.with_suffix('.so.tmp')
        subprocess.run(['g++', '-shared', '-fPIC', *SAN, '-o', str(tmp), *[str(o) for o in objs]], check=True)
        os.replace(tmp, OUT)                # atomic: a process that already mapped the old file keeps it
    return OUT


if __name__ == '__main__':
    import sys
    print(build(force=True, product='--product' in sys.argv))
