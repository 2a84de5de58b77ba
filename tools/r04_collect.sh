#!/bin/bash
# copies the outputs of tools/r04_measure.sh (gpurun_out/r04_*) into profiles/ under their committed names
G=gpurun_out; P=profiles
{ echo "# bench.py on one MI355X, round 4 (tools/r04_measure.sh -> tools/final_measure.sh); default = shipped flags, C1 = 1140, batch 32; bN = N graphs per GPU"; echo "# (the strong-scaling shards of the reference's DataParallel batch of 32 at 2 / 4 / 8 GPUs are b16 / b8 / b4); c180 = max_num_nodes 1800; c5 = 8000 nodes, 64 features, C1 = 1600"; echo "# (c5_draw: nuclei in draw order -- the model re-lists large graphs grid cell by grid cell itself; c5_spatial: already listed that way by the caller)"; grep -E "graphs/s" $G/r04_configurations_raw.txt; } > $P/r04_configurations.txt
cp $G/r04_c3_trace.txt $P/r04_bench_c3_shipped_kernel_trace.txt; cp $G/r04_b4_trace.txt $P/r04_bench_c3_batch4_kernel_trace.txt; cp $G/r04_c5_trace.txt $P/r04_c5_draw_order_kernel_trace.txt
{ echo "# GPU idle time inside steady-state steps under rocprofv3 --kernel-trace (tools/prof_gaps.sh, profiles/gaps_rocpd.py): batch 32, then batch 4"; echo "# (tracing costs the host ~3 us per launch, ~230 launches per step at batch 4)"; cat $G/r04_c3_gaps.txt; cat $G/r04_b4_gaps.txt; } > $P/r04_step_gaps.txt
cp $G/r04_bench_c3_pmc_sq.txt $G/r04_bench_c3_pmc_traffic.txt $G/r04_counters.json $G/r04_traffic.json $G/r04_gemm_calls_by_shape.txt $G/r04_gemm_tail_split.txt $P/
grep -v amdgpu $G/r04_eval.txt > $P/r04_eval_throughput.txt; grep -v amdgpu $G/r04_gemm_standalone.txt > $P/r04_gemm_standalone.txt; grep -v amdgpu $G/r04_fp64_report.txt > $P/r04_gradients_vs_reference_fp64.txt; tail -1 $G/r04_adj_fused.json > $P/r04_bench_adj_fused_line.json
tail -1 $G/r04_default.json > $P/r04_bench_default.json; tail -1 $G/r04_c5_draw.json > $P/r04_c5_line.json; tail -1 $G/r04_b4.json > $P/r04_bench_b4_line.json
python - <<'PY'
import json,sys
sys.path.insert(0,'.')
from bench import source_hash
c=json.load(open('profiles/r04_counters.json')); t=json.load(open('profiles/r04_traffic.json'))
print('tree hash', source_hash()[:12], 'counters', c['source_sha256'][:12], 'traffic', t['source_sha256'][:12], c['gemm_128x128']['mfma_busy'])
PY
tail -10 $P/r04_configurations.txt; tail -3 $P/r04_step_gaps.txt | cut -c1-110
