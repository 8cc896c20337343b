#!/bin/bash
# closing profile of round 4 on ONE box: the default bench line, then the rocprofv3 kernel stats of the same command and the two counter passes of its
# child command (tools/r4_profile_only.sh) -- so that the committed line and the committed summaries come from the same box and tree
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 400 python bench.py > gpurun_out/r04_bench_s2.json 2> gpurun_out/r04_bench_s2.err
bash tools/r4_profile_only.sh > gpurun_out/r04_profile_only.log 2>&1
cut -c1-170 gpurun_out/r04_bench_s2.json; head -8 gpurun_out/r04_stats.txt
