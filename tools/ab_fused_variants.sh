#!/bin/bash
# The fused backward+Adam kernel (BASELINE.json configs[3]) under occupancy caps / phase-B depths, alternating processes on ONE box:
# libfgs_hip_fused_<name>.so built by tools/build_variant.sh fused_<name> preprocess_backward.hip -DFGS_FUSED_WAVES=.. -DFGS_FUSED_UNROLL=..
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2; do for lib in libfgs_hip.so $(cd $P; ls libfgs_hip_fused_*.so); do
  echo -n "$lib $r: "; FGS_HIP_LIBRARY=$P/$lib timeout 120 python tools/fused_times.py 2>/dev/null | tail -1
done; done
