#!/bin/bash
# A/B of two builds of the library on ONE box: faster-gaussian-splatting_amd/libfgs_hip_ref.so (a build of another commit) vs the current
# one, alternating bench.py runs; prints ms/step and the stage times named on the command line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for w in ref new; do
  if [ $w = ref ]; then export FGS_HIP_LIBRARY=$PWD/faster-gaussian-splatting_amd/libfgs_hip_ref.so; else unset FGS_HIP_LIBRARY; fi
  python bench.py --steps 16 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ablib_${w}_$r.json
done; done
python - "$@" <<'PY'
import json, sys
keys = sys.argv[1:] or ['blend_backward']
for r in (1, 2):
    for w in ('ref', 'new'):
        d = json.loads(open(f'gpurun_out/ablib_{w}_{r}.json').read())
        print(w, r, round(d['ms_per_step'], 3), {k: round(v, 4) for k, v in d['stage_ms_per_step'].items() if k in keys})
PY
