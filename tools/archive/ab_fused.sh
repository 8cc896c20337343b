#!/bin/bash
# A/B of builds of the library on the fused iteration, alternating processes on ONE box. usage: bash tools/ab_fused.sh libA.so libB.so ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2; do for lib in libfgs_hip.so "$@"; do
  FGS_HIP_LIBRARY=$P/$lib python tools/fused_times.py 2>/dev/null | sed "s/^/$lib $r  /"
done; done
