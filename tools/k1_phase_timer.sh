#!/bin/bash
# Where does K1 (preprocess) spend its time? Builds a SEPARATE library with cycle-counter probes between the phases of preprocess_body
# (-DFGS_K1_PHASE_TIMER, csrc/preprocess.hip) -- the product library is untouched -- and prints each phase's share of the wave-cycles.
# usage: bash tools/k1_phase_timer.sh build   (here: cross-compiles)      bash tools/k1_phase_timer.sh run   (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/faster-gaussian-splatting_amd/csrc; LIB=$R/faster-gaussian-splatting_amd/libfgs_hip_k1timer.so
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include -ffp-contract=off -DFGS_K1_PHASE_TIMER \
      -c $C/preprocess.hip -o $C/_build/preprocess_k1timer.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $(ls $C/_build/*.o | grep -v "/preprocess.o\|preprocess_k1timer.o\|/bb_\|/bf_") $C/_build/preprocess_k1timer.o
  ls -la $LIB | awk '{print $5, $9}'
else
  FGS_HIP_LIBRARY=$LIB python $R/tools/k1_phase_timer.py
fi
