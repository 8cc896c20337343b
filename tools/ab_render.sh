#!/bin/bash
# A/B of library builds on ONE box for the forward-only frame (BASELINE.json configs[1]) and the training iteration: alternating processes.
# usage: bash tools/ab_render.sh libA.so [libB.so ...]      (libfgs_hip.so is always the first contestant)
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2 3; do for lib in libfgs_hip.so "$@"; do
  FGS_HIP_LIBRARY=$P/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-trained-like --blocks 2 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', $r, 'train ms', round(d['ms_per_step'],3), 'render ms', round(d['render']['ms_per_frame'],4), 'fused ms', round(d['fused']['ms_per_step'],3), 'layered ms', round(d['layered_scene']['ms_per_step'],3))"
done; done
