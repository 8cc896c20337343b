#!/bin/bash
# the full GPU suite with the tolerance log, smoke() and one default bench line (no rocprofv3 passes): gpurun_out/r04_*
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
rm -f gpurun_out/tol.log
FGS_TOL_LOG=$R/gpurun_out/tol.log timeout 1100 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04_gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gpu_tests.txt
python tools/summarize_tol_log.py gpurun_out/tol.log > gpurun_out/r04_gpu_tolerance_slack.txt 2>&1
rm -f gpurun_out/tol.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.txt 2>&1
timeout 400 python bench.py > gpurun_out/r04_bench_line_final.json 2> gpurun_out/r04_bench_final.err
tail -3 gpurun_out/r04_gpu_tests.txt; tail -1 gpurun_out/r04_smoke.txt; cut -c1-400 gpurun_out/r04_bench_line_final.json
