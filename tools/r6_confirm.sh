#!/bin/bash
# Short confirmation (tag = $1) of a host-side change on one MI355X box -> gpurun_out/${1:-r06g}_*: the full GPU suite (stops at the first failure), smoke(), the headline line
# without extras / CPU baseline / counter passes.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$PWD/gpurun_out; T=${1:-r06g}
timeout 330 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/${T}_gpu_tests.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_smoke.txt 2>&1
timeout 150 python bench.py --no-extras --no-cpu-baseline --no-pmc > $O/${T}_bench_s2.json 2> $O/${T}_bench_s2.err
grep -E "passed|failed|rc " $O/${T}_gpu_tests.txt | tail -2; tail -1 $O/${T}_smoke.txt; cut -c1-400 $O/${T}_bench_s2.json
