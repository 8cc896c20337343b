#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=gpurun_out
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace -d $R/$O/seq -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > $R/$O/seq.log 2>&1 )
db=$(find $O/seq -name '*.db' | head -1); python tools/kernel_sequence.py $db 140 > $O/seq_unfused.txt 2>&1; find $O/seq -name '*.db' -delete
python bench.py --scene S0 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-extras > $O/bench_s0.json 2>/dev/null
python - <<'PY' > $O/cpu_profile.txt 2>&1
import cProfile, pstats, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from harness import trainer as T
from harness.scenes import make_s0
params, view = make_s0()
dev = torch.device('cuda:0')
g = T.Gaussians(params, dev); g.training_setup(training_cameras_extent=5.0)
v = view.to(dev); tgt = T.render_image_benchmark(g, v).clone() * 0.9
for i in range(50): T.training_iteration(g, v, tgt, i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(300): T.training_iteration(g, v, tgt, 50 + i)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
PY
tail -70 $O/seq_unfused.txt; python -c "
import json; d=json.loads(open('$O/bench_s0.json').read()); print('S0 ms/step', d['ms_per_step'], sum(d['stage_ms_per_step'].values()))"
head -75 $O/cpu_profile.txt
