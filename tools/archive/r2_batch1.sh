#!/bin/bash
# Round-2 GPU batch 1: full GPU test suite, K11 / fused A-B, VALU micro-benchmark, bench line, kernel stats.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $O/b1_pytest.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/valu_rate.hip -o /tmp/valu_rate > $O/b1_valu_build.log 2>&1 && timeout 120 /tmp/valu_rate > $O/b1_valu_rate.txt 2>&1
timeout 300 python tools/ab_backward.py 3000000 2,3 > $O/b1_ab_backward.txt 2>&1
timeout 400 python tools/scene_sensitivity.py > $O/b1_scene_sensitivity.txt 2>&1
timeout 300 python tools/ab_fused.py > $O/b1_ab_fused.txt 2>&1
timeout 600 python bench.py > $O/b1_bench_s2.json 2> $O/b1_bench_s2.err
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b1_stats -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $R/$O/b1_stats.log 2>&1
cd $R
db=$(find $O/b1_stats -name '*.db' | head -1); python profiles/summarize_rocprof.py stats $db > $O/b1_kernel_stats.txt 2>&1
find $O/b1_stats -name '*.db' -delete
tail -5 $O/b1_pytest.log; cat $O/b1_ab_backward.txt $O/b1_ab_fused.txt; head -c 1500 $O/b1_bench_s2.json
