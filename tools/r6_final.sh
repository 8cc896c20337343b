#!/bin/bash
# Round-6 closing evidence on ONE MI355X box (final tree) -> gpurun_out/r06f_*: the full GPU suite with the tolerance log, smoke(), the default bench line,
# the rocprofv3 kernel stats of the same command and the two counter passes of its child command, S1 / S3 lines, the sharded emulation, the multi-rank
# branch on one device, and (FGS_R6_MCMC=1) the bench line of a 1.5 M model trained from scratch under the MCMC policy.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out; T=r06f
rm -f $O/tol.log
FGS_TOL_LOG=$O/tol.log timeout 1300 python -m pytest tests -m gpu -q --durations=12 > $O/${T}_gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/${T}_gpu_tests.txt
python tools/summarize_tol_log.py $O/tol.log > $O/${T}_gpu_tolerance_slack.txt 2>&1; rm -f $O/tol.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_smoke.txt 2>&1
timeout 600 python bench.py > $O/${T}_bench_s2.json 2> $O/${T}_bench_s2.err
( cd /tmp; export TMPDIR=/tmp
  B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-trained-like"
  P="python $R/bench.py --steps 3 --warmup 1 --no-extras --blocks 1 --no-cpu-baseline --no-pmc --pmc-child"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_stats -o bench -- $B > $O/${T}_stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU -d $O/${T}_fetch -o bench -- $P > $O/${T}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAVES -d $O/${T}_write -o bench -- $P > $O/${T}_write.log 2>&1 )
python profiles/summarize_rocprof.py stats $(find $O/${T}_stats -name '*.db' | head -1) > $O/${T}_stats.txt
( cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $O/${T}_seq -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-extras --blocks 1 --no-cpu-baseline --no-pmc > $O/${T}_seq.log 2>&1 )
python tools/kernel_sequence.py $(find $O/${T}_seq -name '*.db' | head -1) 110 > $O/${T}_kernel_sequence.txt 2>&1; find $O/${T}_seq -name '*.db' -delete
for k in fetch write; do python profiles/summarize_rocprof.py pmc $(find $O/${T}_$k -name '*.db' | head -1) > $O/${T}_$k.txt; done
find $O/${T}_stats $O/${T}_fetch $O/${T}_write -name '*.db' -delete
timeout 300 python bench.py --scene S1 --no-cpu-baseline --no-pmc --no-trained-like > $O/${T}_bench_s1.json 2>/dev/null
timeout 300 python bench.py --scene S3 --no-cpu-baseline --no-pmc --no-trained-like > $O/${T}_bench_s3.json 2>/dev/null
timeout 300 python tools/sharded_emulation.py 2>&1 | grep -v amdgpu.ids > $O/${T}_sharded_emulation.txt
timeout 300 python bench.py --gpus 2 --shared-device --no-cpu-baseline --no-pmc --steps 8 > $O/${T}_bench_2ranks_shared_device.json 2> $O/${T}_bench_2ranks.err
if [ "${FGS_R6_MCMC:-0}" = 1 ]; then
  timeout 400 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > $O/${T}_train_full_mcmc.json 2> /dev/null
  timeout 400 python bench.py --ply /tmp/mcmc.ply --no-cpu-baseline --no-extras --blocks 3 > $O/${T}_trained_mcmc_bench.json 2> $O/${T}_trained_mcmc_bench.err
  rm -f /tmp/mcmc.ply
fi
grep -E "passed|failed|rc " $O/${T}_gpu_tests.txt | tail -2; tail -1 $O/${T}_smoke.txt; cut -c1-200 $O/${T}_bench_s2.json; head -12 $O/${T}_stats.txt; tail -4 $O/${T}_sharded_emulation.txt
