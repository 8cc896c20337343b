#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/b2_pytest.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/valu_rate.hip -o /tmp/valu_rate > $O/b2_valu_build.log 2>&1 && timeout 120 /tmp/valu_rate > $O/b2_valu_rate.txt 2>&1
R=$PWD
( cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $R/$O/b2_valu_pmc -o v -- /tmp/valu_rate > $R/$O/b2_valu_pmc.log 2>&1 )
db=$(find $O/b2_valu_pmc -name '*.db' | head -1); python profiles/summarize_rocprof.py pmc $db > $O/b2_valu_pmc.txt 2>&1; find $O/b2_valu_pmc -name '*.db' -delete
timeout 300 python tools/ab_backward.py 3000000 2,3 > $O/b2_ab_backward.txt 2>&1
timeout 400 python tools/scene_sensitivity.py > $O/b2_scene_sensitivity.txt 2>&1
timeout 600 python bench.py > $O/b2_bench_s2.json 2> $O/b2_bench_s2.err
tail -30 $O/b2_pytest.log; cat $O/b2_ab_backward.txt; grep -v amdgpu $O/b2_scene_sensitivity.txt
