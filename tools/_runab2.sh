cd $GRAFT_REPO_ROOT
for r in 1 2; do
for w in old new; do
  if [ $w = old ]; then d=tools/_ab_old; else d=.; fi
  (cd $d && python bench.py --steps 16 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/ab2_${w}_$r.json
done; done
python - <<'PY'
import json
for r in (1,2):
  for w in ('old','new'):
    d=json.loads(open(f'gpurun_out/ab2_{w}_{r}.json').read())
    print(w, r, round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if k in ('preprocess','blend_backward','preprocess_backward','sh_rest_backward','adam','create_instances')})
PY
