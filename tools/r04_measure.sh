#!/bin/bash
# Round-4 measurements on the GPU box, one call: tools/r04_measure.sh  -> gpurun_out/r04_* (copied into profiles/ afterwards)
# 1 configurations table  2 kernel traces + step gaps (32 and 4 graphs per GPU)  3 SQ / TCC counter passes stamped with the source
# hash  4 C5 in draw order  5 per-shape GEMM times
export TMPDIR=/tmp
R=$(pwd)
B="python $R/bench.py"
bash tools/final_measure.sh r04 > gpurun_out/r04_configurations_raw.txt 2>&1
bash tools/prof_step.sh r04_c3 > /dev/null 2>&1
bash tools/prof_step.sh r04_b4 --batch 4 > /dev/null 2>&1
bash tools/prof_gaps.sh r04_c3 > /dev/null 2>&1
bash tools/prof_gaps.sh r04_b4 --batch 4 > /dev/null 2>&1
PMC_PASSES=3 bash tools/pmc_run.sh r04_sq "" -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
grep -E "^kernel|^k_gemm_f32<2, 2, 2, 2|^k_spmm_wide|^k_jku_bwd<20|^k_jku_fwd<20|^k_gemm_f32<2, 2, 1, 1|^k_gemm_f32<4, 1, 1, 1|^k_gemm_f32_shortk<2, 2, 2, 2|^k_sage_wide_fwd|^k_gemm_fixup" gpurun_out/r04_sq_pmc.txt > gpurun_out/r04_bench_c3_pmc_sq.txt
python profiles/make_counters_json.py gpurun_out/r04_bench_c3_pmc_sq.txt > gpurun_out/r04_counters.json
bash tools/pmc_tcc.sh r04_tcc -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python profiles/make_traffic_json.py gpurun_out/r04_tcc_FETCH_SIZE gpurun_out/r04_tcc_WRITE_SIZE > gpurun_out/r04_traffic.json
python profiles/summarize_pmc.py gpurun_out/r04_tcc_FETCH_SIZE gpurun_out/r04_tcc_WRITE_SIZE gpurun_out/r04_tcc_TCC --match "k_gemm_f32<2, 2, 2, 2" > gpurun_out/r04_bench_c3_pmc_traffic.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/r04_tcc_FETCH_SIZE gpurun_out/r04_tcc_WRITE_SIZE gpurun_out/r04_tcc_TCC --match "k_spmm_wide" >> gpurun_out/r04_bench_c3_pmc_traffic.txt 2>&1
rm -rf gpurun_out/r04_tcc_FETCH_SIZE gpurun_out/r04_tcc_WRITE_SIZE gpurun_out/r04_tcc_TCC
NSTEPS=8 bash tools/prof_step.sh r04_c5 --nodes 8000 --feat 64 --maxn 16000 --steps 6 --warmup 2 --pool 2 > /dev/null 2>&1
CGC_NATIVE=0 python tools/gemm_time_shapes.py 32 > gpurun_out/r04_gemm_calls_by_shape.txt 2>&1
python tools/gemm_tail_bench.py 4 8 16 32 > gpurun_out/r04_gemm_tail_split.txt 2>&1
python tools/eval_bench.py 32 160 > gpurun_out/r04_eval.txt 2>&1
python tools/eval_bench.py 4 64 >> gpurun_out/r04_eval.txt 2>&1
CGC_ADJ_FUSED=1 python bench.py --no-cpu-baseline > gpurun_out/r04_adj_fused.json 2>/dev/null
python tools/gemm_quick.py > gpurun_out/r04_gemm_standalone.txt 2>&1
python tools/golden_fp64_report.py > gpurun_out/r04_fp64_report.txt 2>&1
ls gpurun_out | grep r04_ | head -80
