#!/bin/bash
# Round-3 evidence run on ONE MI355X box: the full GPU suite with the tolerance log, then the profile round (tools/profile_round.sh r03),
# the blend-kernel timelines on both scenes and the in-process A/B of K10's mappings. Everything lands in gpurun_out/r03_*.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
rm -f gpurun_out/tol.log
FGS_TOL_LOG=$R/gpurun_out/tol.log timeout 900 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r03_gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03_gpu_tests.txt
python tools/summarize_tol_log.py gpurun_out/tol.log > gpurun_out/r03_gpu_tolerance_slack.txt 2>&1
timeout 1200 bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
timeout 200 bash tools/k10_timeline.sh run > gpurun_out/r03_k10_timeline.txt 2>&1
timeout 200 bash tools/k11_timeline.sh run > gpurun_out/r03_k11_timeline.txt 2>&1
timeout 300 python tools/ab_tile_plan.py > gpurun_out/r03_ab_tile_plan.txt 2>&1
echo done > gpurun_out/r03_final.done
