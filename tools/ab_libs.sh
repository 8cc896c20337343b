#!/bin/bash
# A/B of several builds of the library on ONE box (generalisation of ab_lib.sh): the current library plus the named variants
# (faster-gaussian-splatting_amd/<name>), alternating bench runs; prints ms/step and the stages named after "--".
# usage: bash tools/ab_libs.sh libA.so libB.so -- blend_backward stage_pixels
cd ${GRAFT_REPO_ROOT:-/root/repo}
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2; do for lib in libfgs_hip.so "${libs[@]}"; do
  FGS_HIP_LIBRARY=$P/$lib python bench.py --steps 16 --warmup 3 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); ks=sys.argv[1:]; print('$lib', $r, round(d['ms_per_step'],3), {k: round(v,4) for k,v in d['stage_ms_per_step'].items() if k in ks})" "$@"
done; done
