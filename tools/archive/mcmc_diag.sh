#!/bin/bash
# Diagnosis of K10 / K11 on a model trained from scratch under the MCMC policy (1.5 M Gaussians): pair statistics, K11's item timeline, ablations.
# needs tools/pair_stats.sh build and tools/k11_timeline.sh build; run on the GPU box
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > /dev/null 2>&1
ls -la /tmp/mcmc.ply
export FGS_PLY=/tmp/mcmc.ply
FGS_PAIR_STATS_ONLY_PLY=1 bash tools/pair_stats.sh run 2>&1 | grep -v amdgpu.ids > gpurun_out/mcmc_pair_stats.txt
FGS_HIP_LIBRARY=$PWD/faster-gaussian-splatting_amd/libfgs_hip_k11timeline.so python tools/k11_timeline.py 0.0 2>&1 | grep -v amdgpu.ids > gpurun_out/mcmc_k11_timeline.txt
for ab in 0 1 2; do echo "== K11 ablation bits $ab (1: no atomics, 2: no step loop)"; FGS_ABLATE=$ab python tools/ab_k11m.py 3 2>&1 | grep -A1 "^PLY"; done > gpurun_out/mcmc_k11_ablate.txt
cat gpurun_out/mcmc_pair_stats.txt gpurun_out/mcmc_k11_timeline.txt gpurun_out/mcmc_k11_ablate.txt
