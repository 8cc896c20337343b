#!/bin/bash
# Round-4 GPU batch 4: the driver's bench line (with the trained_like block and the layered scene's counter passes), and the from-scratch run on a denser mosaic.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 600 python bench.py > $O/r04_bench_s2.json 2> $O/r04_bench_s2.err ) 2> $O/r04_bench_s2.time
timeout 600 python tools/train_full.py --gt ${GT:-2500000} --max-gaussians 5000000 --max-seconds 400 > $O/r04_train_full.json 2> $O/r04_train_full.err
echo done > $O/r04_batch4.done
