#!/bin/bash
# rocprofv3 passes behind profiles/: kernel stats, then HBM / VALU counters in separate --pmc runs (MI355X_MICROARCH.md, HBM section).
# usage (on the GPU box): bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*.txt + bench lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o bench -- $B > $R/gpurun_out/${TAG}_stats.log 2>&1
# counter passes: the child command of bench.py's own live counters (headline training steps + fused steps; no layered / trained-like extras, whose
# launches at other sizes would be averaged into the per-dispatch totals)
P="python $R/bench.py --steps 3 --warmup 1 --no-extras --blocks 1 --no-cpu-baseline --no-pmc --pmc-child"
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU -d $R/gpurun_out/${TAG}_fetch -o bench -- $P > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAVES -d $R/gpurun_out/${TAG}_write -o bench -- $P > $R/gpurun_out/${TAG}_write.log 2>&1
S="python $R/bench.py --gpus 1 --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-pmc --dp-mode sharded --force-dp"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_sharded_stats -o bench -- $S > $R/gpurun_out/${TAG}_sharded_stats.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value $R/tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > $R/gpurun_out/${TAG}_valu_rate.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate > $R/gpurun_out/${TAG}_atomic_rate.txt 2>&1
cd $R
for k in stats sharded_stats; do db=$(find gpurun_out/${TAG}_$k -name '*.db' | head -1); python profiles/summarize_rocprof.py stats $db > gpurun_out/${TAG}_$k.txt; done
for k in fetch write; do db=$(find gpurun_out/${TAG}_$k -name '*.db' | head -1); python profiles/summarize_rocprof.py pmc $db > gpurun_out/${TAG}_$k.txt; done
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sharded_stats -name '*.db' -delete
python bench.py > gpurun_out/${TAG}_bench_s2.json 2> gpurun_out/${TAG}_bench_s2.err
python bench.py --scene S1 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_s1.json 2>/dev/null
python bench.py --scene S3 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_s3.json 2>/dev/null
python tools/ablate_k11.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_k11_ablation.txt
python tools/time_densify.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_time_densify.txt
head -14 gpurun_out/${TAG}_stats.txt
