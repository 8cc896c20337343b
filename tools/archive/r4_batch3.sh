#!/bin/bash
# Round-4 GPU batch 3: K11 variant 4 (final schedule): A/B against variant 3 on S2 / layered / a trained export, and its ablations.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
PLY=/tmp/trained_full.ply
timeout 600 python tools/train_full.py --gt 1200000 --iters 8000 --eval-at 8000 --save-ply $PLY > $O/r04_train_short.json 2> $O/r04_train_short.err
FGS_PLY=$PLY timeout 400 python tools/ab_k11m.py 2>&1 | grep -v amdgpu > $O/r04_ab_k11m.txt
for ab in 4 12; do echo "== ablate $ab (4: no matrix instructions / write-out, 8: no pair arithmetic)"; FGS_ABLATE=$ab timeout 300 python tools/ab_k11m.py 4 2>&1 | grep -v "amdgpu\|diff"; done > $O/r04_k11m_ablation.txt
echo done > $O/r04_batch3.done
