#!/bin/bash
# K11's accumulators as per-Gaussian records (this tree) against the planar form (libfgs_hip_ref.so = a build of the commit before): parity subset, then
# stage times on S2 / layered (tools/ab_two_libs.sh) and on a model trained under the MCMC policy (bench.py --ply).
cd ${GRAFT_REPO_ROOT:-/root/repo}; P=$PWD/faster-gaussian-splatting_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py tests/test_gpu_multi_process.py tests/test_gpu_rccl.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/rec_tests.txt
cat gpurun_out/rec_tests.txt
bash tools/ab_two_libs.sh blend_backward preprocess_backward adam > gpurun_out/rec_ab_s2.txt 2>&1; cat gpurun_out/rec_ab_s2.txt
timeout 300 python tools/train_full.py --policy mcmc --max-primitives 1500000 --gt 2500000 --save-ply /tmp/mcmc.ply --eval-at 30000 > /dev/null 2>&1
for r in 1 2; do for w in ref new; do
  if [ $w = ref ]; then export FGS_HIP_LIBRARY=$P/libfgs_hip_ref.so; else unset FGS_HIP_LIBRARY; fi
  python bench.py --ply /tmp/mcmc.ply --steps 16 --warmup 3 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', $r, 'MCMC-trained 1.5 M:', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', {k: round(v,4) for k,v in d['stage_ms_per_step'].items() if k in ('blend_backward','preprocess_backward','adam','blend_forward')})"
done; done > gpurun_out/rec_ab_mcmc.txt 2>&1; cat gpurun_out/rec_ab_mcmc.txt
