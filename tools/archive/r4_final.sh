#!/bin/bash
# Round-4 evidence run on ONE MI355X box: the full GPU suite with the tolerance log, then the profile round (tools/profile_round.sh r04).
# Everything lands in gpurun_out/r04_*.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
rm -f gpurun_out/tol.log
FGS_TOL_LOG=$R/gpurun_out/tol.log timeout 900 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04_gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gpu_tests.txt
python tools/summarize_tol_log.py gpurun_out/tol.log > gpurun_out/r04_gpu_tolerance_slack.txt 2>&1
rm -f gpurun_out/tol.log
timeout 1200 bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
echo done > gpurun_out/r04_final.done
