#!/bin/bash
# Where does a wave of K11's lane = pixel variant spend its cycles? Separate library with cycle-counter probes between the phases
# (-DFGS_K11M_PHASES, csrc/blend_backward.hip; the product library is untouched).
# usage: bash tools/k11m_phases.sh build   (here)      bash tools/k11m_phases.sh run   (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/faster-gaussian-splatting_amd/csrc; LIB=$R/faster-gaussian-splatting_amd/libfgs_hip_k11mphases.so
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include -Xclang -target-feature -Xclang -packed-fp32-ops \
      -DFGS_K11M_PHASES -c $C/blend_backward.hip -o $C/_build/bb_k11mphases.o 2>&1 | grep -v packed-fp32 || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $(ls $C/_build/*.o | grep -v "blend_backward.o\|/bb_\|/bf_\|k1timer") $C/_build/bb_k11mphases.o
  ls -la $LIB | awk '{print $5, $9}'
else
  FGS_HIP_LIBRARY=$LIB python $R/tools/k11m_phases.py
fi
