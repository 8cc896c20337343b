#!/bin/bash
# A/B of two builds of the library on ONE box, alternating processes: faster-gaussian-splatting_amd/libfgs_hip_ref.so (a build of the commit before)
# against the current one; S2 and the layered scene (tools/stage_times.py). usage: bash tools/ab_two_libs.sh [stage ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2 3; do for w in ref new; do
  if [ $w = ref ]; then export FGS_HIP_LIBRARY=$P/libfgs_hip_ref.so; else unset FGS_HIP_LIBRARY; fi
  python tools/stage_times.py "$@" 2>/dev/null | sed "s/^/$w $r  /"
done; done
