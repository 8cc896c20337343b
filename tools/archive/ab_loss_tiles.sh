#!/bin/bash
# A/B of loss-kernel tile shapes on ONE box: libfgs_hip_ref.so (HEAD: 32x16, one LDS read per FMA) vs builds of the register-blocked
# kernels with -DFGS_LOSS_TILE_W/H (faster-gaussian-splatting_amd/libfgs_hip_loss_WxH.so); prints the l1_dssim_loss stage per run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/faster-gaussian-splatting_amd
for r in 1 2; do for lib in libfgs_hip_ref.so "$@"; do
  FGS_HIP_LIBRARY=$P/$lib python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', $r, round(d['ms_per_step'],3), round(d['stage_ms_per_step']['l1_dssim_loss'],4))"
done; done
