#!/bin/bash
# Round-4 GPU batch 5: the TRAINED 2.9 M-Gaussian scene of round 3 (tools/train_demo.py --n 6000000: perturbed-subsample initialisation, 3000 iterations)
# with the evidence VERDICT r3 missed: pair statistics of K10 / K11, counter passes, kernel trace, bench line with live counters.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
PLY=/tmp/trained3m.ply
timeout 300 python tools/train_demo.py --n 6000000 --iters 3000 --save-ply $PLY > $O/r04_trained3m_training.json 2> $O/r04_trained3m_training.err
FGS_PLY=$PLY FGS_PAIR_STATS_ONLY_PLY=1 timeout 300 bash tools/pair_stats.sh run > $O/r04_trained3m_pair_efficiency.txt 2>&1
timeout 500 python bench.py --ply $PLY --no-cpu-baseline --no-extras --blocks 3 > $O/r04_trained3m_bench.json 2> $O/r04_trained3m_bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --ply $PLY --no-cpu-baseline --no-extras --no-pmc --blocks 1 --steps 8 --warmup 2"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/r04_stats -o t -- $B > $O/r04_trained3m_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU -d /tmp/r04_fetch -o t -- $B >> $O/r04_trained3m_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAVES -d /tmp/r04_write -o t -- $B >> $O/r04_trained3m_prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/r04_sq -o t -- $B >> $O/r04_trained3m_prof.log 2>&1
python $R/profiles/summarize_rocprof.py stats $(find /tmp/r04_stats -name '*.db' | head -1) > $O/r04_trained3m_kernel_stats.txt 2>&1
for k in fetch write sq; do python $R/profiles/summarize_rocprof.py pmc $(find /tmp/r04_$k -name '*.db' | head -1) > $O/r04_trained3m_pmc_$k.txt 2>&1; done
echo done > $O/r04_batch5.done
