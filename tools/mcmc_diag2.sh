#!/bin/bash
# K11 on the MCMC-trained model: product kernel vs no atomics vs the consecutive-record atomics experiment (ablate bit 16)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > /dev/null 2>&1
export FGS_PLY=/tmp/mcmc.ply
for ab in 0 1 16; do echo "== K11 ablation bits $ab (1: no atomics, 16: nine consecutive floats per Gaussian, 7 Gaussians per instruction)"; FGS_ABLATE=$ab python tools/ab_k11m.py 3 2>&1 | grep "^S2\|^layered\|^PLY\|median"; done > gpurun_out/mcmc_k11_atomics.txt
cat gpurun_out/mcmc_k11_atomics.txt
