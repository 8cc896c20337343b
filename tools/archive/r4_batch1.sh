#!/bin/bash
# Round-4 GPU batch 1 (one MI355X box): from-scratch full-schedule training run -> its export as the TRAINED scene for (a) the pair statistics
# of K10 / K11, (b) bench.py --ply with live PMC passes, (c) a rocprofv3 kernel trace. Everything lands in gpurun_out/r04_*.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
PLY=/tmp/trained_full.ply
timeout 420 python tools/train_full.py --max-gaussians 4000000 --max-seconds 300 --save-ply $PLY > $O/r04_train_full.json 2> $O/r04_train_full.err
echo "train_full rc $?" >> $O/r04_train_full.err
if [ ! -s $PLY ]; then   # fallback: the round-3 demo run (perturbed-subsample initialisation) so that the measurements below still have a trained scene
  timeout 200 python tools/train_demo.py --n 6000000 --iters 3000 --save-ply $PLY > $O/r04_train_demo_fallback.json 2>&1
fi
if [ "$1" = train-only ]; then
  timeout 400 python bench.py --ply $PLY --no-cpu-baseline --no-pmc --no-extras --blocks 3 > $O/r04_trained_full_bench.json 2> $O/r04_trained_full_bench.err
  echo done > $O/r04_batch1.done; exit 0
fi
FGS_PLY=$PLY timeout 300 bash tools/pair_stats.sh run > $O/r04_k11_pair_efficiency.txt 2>&1
timeout 400 python bench.py --ply $PLY --no-cpu-baseline --no-pmc --blocks 3 > $O/r04_trained_full_bench.json 2> $O/r04_trained_full_bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --ply $PLY --no-cpu-baseline --no-extras --no-pmc --blocks 1 --steps 8 --warmup 2"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/r04_stats -o t -- $B > $O/r04_trained_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU -d /tmp/r04_fetch -o t -- $B >> $O/r04_trained_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAVES -d /tmp/r04_write -o t -- $B >> $O/r04_trained_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/r04_sq -o t -- $B >> $O/r04_trained_prof.log 2>&1
python $R/profiles/summarize_rocprof.py stats $(find /tmp/r04_stats -name '*.db' | head -1) > $O/r04_trained_full_kernel_stats.txt 2>&1
for k in fetch write sq; do python $R/profiles/summarize_rocprof.py pmc $(find /tmp/r04_$k -name '*.db' | head -1) > $O/r04_trained_full_pmc_$k.txt 2>&1; done
echo done > $O/r04_batch1.done
