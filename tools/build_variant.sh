#!/bin/bash
# Builds faster-gaussian-splatting_amd/libfgs_hip_<name>.so: the current objects with ONE source recompiled with extra flags (A/B and ablation builds).
# usage: bash tools/build_variant.sh <name> <source.hip> <flags...>       e.g.  bash tools/build_variant.sh k5ab1 binning.hip -DFGS_K5_ABLATE=1
set -e
cd "$(dirname "$0")/../faster-gaussian-splatting_amd/csrc"
name=$1; src=$2; shift 2
make -s -j8 >/dev/null
extra=""
case $src in
  preprocess.hip) extra="-ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops";;
  binning.hip|preprocess_backward.hip|aux_ops.hip|densify.hip) extra="-ffp-contract=off";;
  blend_backward.hip|blend_forward.hip|loss.hip|radix_sort.hip) extra="-Xclang -target-feature -Xclang -packed-fp32-ops";;
esac
obj=/tmp/fgs_variant_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include -Wall -Wno-unused-function $extra "$@" -c $src -o $obj
objs=""
for o in api preprocess binning blend_forward blend_backward preprocess_backward selftest loss aux_ops shard_exchange radix_sort densify; do
  if [ "$o.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs _build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfgs_hip_$name.so $objs
echo built libfgs_hip_$name.so
