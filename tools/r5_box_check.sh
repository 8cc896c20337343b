#!/bin/bash
# Is a slow Adam the box or the build? The headline line from the product and from the dev library (old host path: atomics) on the SAME box, plus the box's
# device-to-device copy rate -> gpurun_out/r05h_*
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$PWD/gpurun_out; T=r05h
timeout 120 python - > $O/${T}_copy_rate.txt 2>&1 <<'PY'
import torch, time
a = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'd2d copy of 1 GiB: {ms:.3f} ms = {2 * (1 << 30) / ms / 1e6:.0f} GB/s (read + write)')
PY
B="python bench.py --no-extras --no-cpu-baseline --no-pmc --blocks 3"
timeout 100 $B > $O/${T}_product.json 2> $O/${T}_product.err
FGS_HIP_LIBRARY=$PWD/faster-gaussian-splatting_amd/libfgs_hip_dev.so timeout 100 $B > $O/${T}_dev.json 2> $O/${T}_dev.err
timeout 100 $B > $O/${T}_product2.json 2> $O/${T}_product2.err
cat $O/${T}_copy_rate.txt
python - <<'PY'
import json, os
O = os.path.join(os.getcwd(), 'gpurun_out')
for n in ('product', 'dev', 'product2'):
    try:
        d = json.loads(open(f'{O}/r05h_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value'], 1), [round(x, 3) for x in d['repeatability']['ms_per_step']], 'adam', round(d['stage_ms_per_step']['adam'], 4), 'K12', round(d['stage_ms_per_step']['preprocess_backward'], 4))
    except Exception as e:
        print(n, 'failed', e)
PY
