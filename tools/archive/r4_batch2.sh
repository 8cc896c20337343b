#!/bin/bash
# Round-4 GPU batch 2: from-scratch training on a denser mosaic ground truth (growth past 1 M), then the K11 A/B (systolic vs lane = pixel + matrix cores)
# on S2, the layered scene and the trained export.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
PLY=/tmp/trained_full.ply
timeout 600 python tools/train_full.py --gt ${GT:-1200000} --max-gaussians 5000000 --max-seconds 400 --save-ply $PLY > $O/r04_train_full.json 2> $O/r04_train_full.err
echo "train_full rc $?" >> $O/r04_train_full.err
FGS_PLY=$PLY timeout 400 python tools/ab_k11m.py > $O/r04_ab_k11m.txt 2>&1
echo done > $O/r04_batch2.done
