#!/bin/bash
# the three rocprofv3 passes of tools/profile_round.sh r04 alone (kernel stats of the default bench command; counter passes of its child command)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r04
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc"
P="python $R/bench.py --steps 3 --warmup 1 --no-extras --blocks 1 --no-cpu-baseline --no-pmc --pmc-child"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o bench -- $B > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU -d $R/gpurun_out/${TAG}_fetch -o bench -- $P > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAVES -d $R/gpurun_out/${TAG}_write -o bench -- $P > $R/gpurun_out/${TAG}_write.log 2>&1
cd $R
python profiles/summarize_rocprof.py stats $(find gpurun_out/${TAG}_stats -name '*.db' | head -1) > gpurun_out/${TAG}_stats.txt
for k in fetch write; do python profiles/summarize_rocprof.py pmc $(find gpurun_out/${TAG}_$k -name '*.db' | head -1) > gpurun_out/${TAG}_$k.txt; done
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write -name '*.db' -delete
timeout 120 python -m pytest tests/test_gpu_parity.py -q -k "equal_depth or selftest" 2>&1 | grep -E "passed|failed" > gpurun_out/${TAG}_new_tests.txt
