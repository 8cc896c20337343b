cd $GRAFT_REPO_ROOT
python bench.py --steps 16 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_single.json
for m in zero1 sharded; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 16 --warmup 3 --no-extras --no-cpu-baseline --dp-mode $m 2>/dev/null | tail -1 > gpurun_out/ab_$m.json
done
python - <<'PY'
import json
for m in ('single','zero1','sharded'):
    d=json.loads(open(f'gpurun_out/ab_{m}.json').read())
    print(m, round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})
PY
