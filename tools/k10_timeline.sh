#!/bin/bash
# When do K10's tiles run and for how long? Separate library with a per-tile timestamp probe (-DFGS_K10_TIMELINE, csrc/blend_forward.hip);
# the product library is untouched.   usage: bash tools/k10_timeline.sh build | run
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/faster-gaussian-splatting_amd/csrc; LIB=$R/faster-gaussian-splatting_amd/libfgs_hip_k10timeline.so
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include -Xclang -target-feature -Xclang -packed-fp32-ops \
      -DFGS_K10_TIMELINE -c $C/blend_forward.hip -o $C/_build/bf_timeline.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $(ls $C/_build/*.o | grep -v "blend_forward.o\|/bb_\|/bf_\|k1timer\|variant") $C/_build/bf_timeline.o
  ls -la $LIB | awk '{print $5, $9}'
else
  for shift in ${SHIFTS:-0.0 -3.0}; do FGS_HIP_LIBRARY=$LIB python $R/tools/k10_timeline.py $shift; done
fi
