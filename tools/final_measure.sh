#!/bin/bash
# Round measurements on the GPU box: default line (with CPU baseline), per-rank-batch sweep (the strong-scaling shards of a
# global batch of 32), the proportional-cluster configurations, C5.  -> gpurun_out/<tag>_*.json
tag=$1
python bench.py > gpurun_out/${tag}_default.json 2> gpurun_out/${tag}_default.err
for b in 16 8 4; do python bench.py --batch $b --no-cpu-baseline --steps 40 > gpurun_out/${tag}_b$b.json 2>/dev/null; done
python bench.py --flags plain --no-cpu-baseline > gpurun_out/${tag}_plain.json 2>/dev/null
python bench.py --maxn 1800 --flags plain --no-cpu-baseline --steps 60 > gpurun_out/${tag}_c180_plain.json 2>/dev/null
python bench.py --maxn 1800 --no-cpu-baseline --steps 60 > gpurun_out/${tag}_c180_shipped.json 2>/dev/null
python bench.py --nodes 8000 --feat 64 --maxn 16000 --steps 6 --warmup 2 --pool 2 --no-cpu-baseline --spatial > gpurun_out/${tag}_c5_spatial.json 2>/dev/null
python bench.py --nodes 8000 --feat 64 --maxn 16000 --steps 6 --warmup 2 --pool 2 --no-cpu-baseline > gpurun_out/${tag}_c5_draw.json 2>/dev/null
python bench.py --nodes 8000 --feat 64 --maxn 16000 --steps 6 --warmup 2 --pool 2 --no-cpu-baseline --flags plain --spatial > gpurun_out/${tag}_c5_plain_spatial.json 2>/dev/null
for f in default b16 b8 b4 plain c180_plain c180_shipped c5_spatial c5_draw c5_plain_spatial; do
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${tag}_$f.json').read().strip().splitlines()[-1])
    r, a = d.get('roofline', {}), d.get('roofline_aggregation', {})
    print('%-18s %9.1f graphs/s %8.3f ms/step  gemm frac %s  K4 frac %s  cpu %s   split mode: %s graphs/s %s ms/step (six products %s ms, bf16 pipe %s)   half mode: %s graphs/s %s ms/step (six products %s ms, fp16 pipe %s)' % (
        '$f', d['value'], d['ms_per_step'], r.get('frac'), a.get('frac'), d.get('cpu_baseline', {}).get('value'), d.get('value_split'),
        d.get('ms_per_step_split'), d.get('roofline_split', {}).get('ms_per_step'), d.get('roofline_split', {}).get('frac'),
        d.get('value_half'), d.get('ms_per_step_half'), d.get('roofline_half', {}).get('ms_per_step'), d.get('roofline_half', {}).get('frac')))
except Exception as e:
    print('$f', 'failed', e)
PY
done
