#!/bin/bash
# When do K11's work items run, for how long and where? Builds a SEPARATE library with a per-item timestamp probe (-DFGS_K11_TIMELINE,
# csrc/blend_backward.hip) -- the product library is untouched -- and prints concurrency over time, item durations and load per XCD.
# usage: bash tools/k11_timeline.sh build   (here: cross-compiles)      bash tools/k11_timeline.sh run   (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/faster-gaussian-splatting_amd/csrc; LIB=$R/faster-gaussian-splatting_amd/libfgs_hip_k11timeline.so
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include -Xclang -target-feature -Xclang -packed-fp32-ops \
      -DFGS_K11_TIMELINE -c $C/blend_backward.hip -o $C/_build/bb_timeline.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $(ls $C/_build/*.o | grep -v "blend_backward.o\|/bb_\|/bf_\|k1timer") $C/_build/bb_timeline.o
  ls -la $LIB | awk '{print $5, $9}'
else
  for shift in 0.0 -3.0; do FGS_HIP_LIBRARY=$LIB python $R/tools/k11_timeline.py $shift; done
fi
