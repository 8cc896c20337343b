#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out
export FGS_HIP_LIBRARY=$R/faster-gaussian-splatting_amd/libfgs_hip_dev.so
cd /tmp; export TMPDIR=/tmp
for v in 3 5; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    tag=$(echo $set | tr ' ' '_')
    FGS_BACKWARD_VARIANT=$v timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/k11pmc_${v}_$tag -o b -- python $R/bench.py --steps 3 --warmup 1 --no-extras --blocks 1 --no-cpu-baseline --no-pmc --pmc-child --opacity-shift -3.0 > /dev/null 2>&1
    echo "== variant $v $set"
    python $R/profiles/summarize_rocprof.py pmc $(find $O/k11pmc_${v}_$tag -name '*.db' | head -1) | grep -E "blend_backward_(compact|chained)"
    rm -rf $O/k11pmc_${v}_$tag
  done
done
