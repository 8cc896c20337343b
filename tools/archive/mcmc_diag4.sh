#!/bin/bash
# K10's per-tile timeline on the MCMC-trained model (needs tools/k10_timeline.sh build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > /dev/null 2>&1
FGS_PLY=/tmp/mcmc.ply FGS_HIP_LIBRARY=$PWD/faster-gaussian-splatting_amd/libfgs_hip_k10timeline.so python tools/k10_timeline.py 0.0 2>&1 | grep -v amdgpu.ids > gpurun_out/mcmc_k10_timeline.txt
cat gpurun_out/mcmc_k10_timeline.txt
