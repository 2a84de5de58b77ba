#!/bin/bash
# Kernel trace of the default bench on the GPU box: tools/prof_step.sh <tag> [extra bench args]
# -> gpurun_out/<tag>_trace.txt (per-kernel summary, 13 steps incl. warm-up) and gpurun_out/<tag>_bench.json
tag=$1; shift
export TMPDIR=/tmp
R=$(pwd)
python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
( cd /tmp && rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/${tag}_prof -o ${tag} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/${tag}_prof.log 2>&1 )
db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
python profiles/summarize_rocpd.py $db --steps ${NSTEPS:-13} > gpurun_out/${tag}_trace.txt
rm -rf gpurun_out/${tag}_prof
head -3 gpurun_out/${tag}_trace.txt
cat gpurun_out/${tag}_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_ms'), d.get('roofline_aggregation',{}).get('frac'))"
