#!/bin/bash
# rocprofv3 passes behind profiles/: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc runs (MI355X_MICROARCH.md, HBM section).
# usage (on the GPU box): bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_{stats,fetch,write}.txt + bench lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r01}
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o bench -- $B > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o bench -- $B > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o bench -- $B > $R/gpurun_out/${TAG}_write.log 2>&1
S="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 $R/bench.py --gpus 1 --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-pmc --dp-mode sharded --force-dp"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_sharded_stats -o bench -- $S > $R/gpurun_out/${TAG}_sharded_stats.log 2>&1
cd $R
for k in stats sharded_stats; do db=$(find gpurun_out/${TAG}_$k -name '*.db' | head -1); python profiles/summarize_rocprof.py stats $db > gpurun_out/${TAG}_$k.txt; done
for k in fetch write; do db=$(find gpurun_out/${TAG}_$k -name '*.db' | head -1); python profiles/summarize_rocprof.py pmc $db > gpurun_out/${TAG}_$k.txt; done
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sharded_stats -name '*.db' -delete
python bench.py > gpurun_out/${TAG}_bench_s2.json 2> gpurun_out/${TAG}_bench_s2.err
python bench.py --scene S1 --no-cpu-baseline > gpurun_out/${TAG}_bench_s1.json 2>/dev/null
python bench.py --scene S3 --no-cpu-baseline > gpurun_out/${TAG}_bench_s3.json 2>/dev/null
head -12 gpurun_out/${TAG}_stats.txt
