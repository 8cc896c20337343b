#!/bin/bash
# How much of the work the two blend kernels issue is useful? Builds a SEPARATE library with counting probes (-DFGS_PAIR_STATS in
# csrc/blend_forward.hip / blend_backward.hip; the product library is untouched) and prints, per scene, K11's pipeline steps / steps whose
# contribution block ran / lane-steps that passed the alpha test, and K10's walked (Gaussian, strip) pairs / lanes blended.
# usage: bash tools/pair_stats.sh build   (here: cross-compiles)      [FGS_PLY=trained.ply] bash tools/pair_stats.sh run   (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/faster-gaussian-splatting_amd/csrc; LIB=$R/faster-gaussian-splatting_amd/libfgs_hip_pairstats.so
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include -Xclang -target-feature -Xclang -packed-fp32-ops -DFGS_PAIR_STATS"
  /opt/rocm/bin/hipcc $F -c $C/blend_backward.hip -o $C/_build/bb_pairstats.o
  /opt/rocm/bin/hipcc $F -c $C/blend_forward.hip -o $C/_build/bf_pairstats.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $(ls $C/_build/*.o | grep -v "blend_backward.o\|blend_forward.o\|/bb_\|/bf_\|k1timer") $C/_build/bb_pairstats.o $C/_build/bf_pairstats.o
  ls -la $LIB | awk '{print $5, $9}'
else
  FGS_HIP_LIBRARY=$LIB python $R/tools/pair_stats.py
fi
