#!/bin/bash
# GPU idle time inside steady-state steps: tools/prof_gaps.sh <tag> [bench args]  -> gpurun_out/<tag>_gaps.txt
tag=$1; shift
export TMPDIR=/tmp
R=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/${tag}_prof -o ${tag} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/${tag}_prof.log 2>&1 )
db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
python profiles/gaps_rocpd.py $db 5 > gpurun_out/${tag}_gaps.txt
rm -rf gpurun_out/${tag}_prof
cat gpurun_out/${tag}_gaps.txt
