cd $GRAFT_REPO_ROOT
for r in 1 2; do
for w in 0 1; do
  FGS_ATOMIC_POLICY=$w python bench.py --steps 16 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab3_${w}_$r.json
done; done
python - <<'PY'
import json
for r in (1,2):
  for w in ('0','1'):
    d=json.loads(open(f'gpurun_out/ab3_{w}_{r}.json').read())
    print('policy', w, r, round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if k in ('blend_backward','preprocess_backward','sh_rest_backward')})
PY
