#!/bin/bash
# K10's tile -> workgroup mappings on the MCMC-trained model (tools/ab_tile_plan.py with FGS_PLY)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > /dev/null 2>&1
FGS_PLY=/tmp/mcmc.ply python tools/ab_tile_plan.py 2>&1 | grep -v amdgpu.ids > gpurun_out/mcmc_tile_plan.txt
cat gpurun_out/mcmc_tile_plan.txt
