#!/bin/bash
# Round-4 closing evidence on ONE MI355X box (final tree): the full GPU suite with the tolerance log, smoke(), the default bench line, the sharded
# emulation, and bench lines of two TRAINED models (the round-3 2.9 M export; a 1.5 M model trained from scratch under the MCMC policy). gpurun_out/r04f_*
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out
rm -f $O/tol.log
FGS_TOL_LOG=$O/tol.log timeout 1100 python -m pytest tests -m gpu -q --durations=12 > $O/r04f_gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/r04f_gpu_tests.txt
python tools/summarize_tol_log.py $O/tol.log > $O/r04f_gpu_tolerance_slack.txt 2>&1; rm -f $O/tol.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04f_smoke.txt 2>&1
timeout 400 python bench.py > $O/r04f_bench_s2.json 2> $O/r04f_bench_s2.err
timeout 300 python tools/sharded_emulation.py 2>&1 | grep -v amdgpu.ids > $O/r04f_sharded_emulation.txt
timeout 300 python tools/train_demo.py --n 6000000 --iters 3000 --save-ply /tmp/trained3m.ply > /dev/null 2>&1
timeout 400 python bench.py --ply /tmp/trained3m.ply --no-cpu-baseline --no-extras --blocks 3 > $O/r04f_trained3m_bench.json 2> $O/r04f_trained3m_bench.err
rm -f /tmp/trained3m.ply
timeout 300 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > /dev/null 2>&1
timeout 400 python bench.py --ply /tmp/mcmc.ply --no-cpu-baseline --no-extras --blocks 3 > $O/r04f_trained_mcmc_bench.json 2> $O/r04f_trained_mcmc_bench.err
grep -E "passed|failed|rc " $O/r04f_gpu_tests.txt | tail -2; tail -1 $O/r04f_smoke.txt; cut -c1-160 $O/r04f_bench_s2.json; tail -6 $O/r04f_sharded_emulation.txt; cut -c1-160 $O/r04f_trained3m_bench.json; cut -c1-160 $O/r04f_trained_mcmc_bench.json
