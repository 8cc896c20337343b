#!/bin/bash
# like ab_two_libs.sh with any number of named builds: bash tools/ab_three_libs.sh libA.so libB.so -- stage ...   (the current build runs too)
cd ${GRAFT_REPO_ROOT:-/root/repo}
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2; do for lib in libfgs_hip.so "${libs[@]}"; do
  FGS_HIP_LIBRARY=$P/$lib python tools/stage_times.py "$@" 2>/dev/null | sed "s/^/$lib $r  /"
done; done
