#!/bin/bash
# Round-6 measurements on the GPU box, one call: tools/r06_measure.sh  -> gpurun_out/r06_* (copied into profiles/ afterwards)
# 1 configurations table (default line incl. the split leg)  2 kernel traces + step gaps: 32 graphs exact / split, 4 graphs, with and
# without Python's garbage collector  3 SQ / TCC counter passes of both GEMM modes stamped with the source hash  4 C5  5 per-shape GEMM
# times, the six products stand-alone, K sweep  6 gradient margins vs the reference fp64 fixtures (default and fused adjacency backward)
export TMPDIR=/tmp
R=$(pwd)
B="python $R/bench.py"
bash tools/final_measure.sh r06 > gpurun_out/r06_configurations_raw.txt 2>&1
bash tools/prof_step.sh r06_c3 --no-split-leg > /dev/null 2>&1
bash tools/prof_step.sh r06_c3_split --gemm-mode 1 --no-split-leg > /dev/null 2>&1
bash tools/prof_step.sh r06_c3_half --gemm-mode 2 --no-split-leg > /dev/null 2>&1
bash tools/prof_step.sh r06_b4 --batch 4 --no-split-leg > /dev/null 2>&1
bash tools/prof_gaps.sh r06_c3 --no-split-leg > /dev/null 2>&1
bash tools/prof_gaps.sh r06_c3_nogc --no-split-leg --no-gc > /dev/null 2>&1
bash tools/prof_gaps.sh r06_b4 --batch 4 --no-split-leg > /dev/null 2>&1
PMC_PASSES=3 bash tools/pmc_run.sh r06_sq "" -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-split-leg > /dev/null 2>&1
PMC_PASSES=3 bash tools/pmc_run.sh r06_sqs "" -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-split-leg --gemm-mode 1 > /dev/null 2>&1
PMC_PASSES=3 bash tools/pmc_run.sh r06_sqh "" -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-split-leg --gemm-mode 2 > /dev/null 2>&1
K='^kernel|^k_gemm_f32<2, 2, 2, 2|^k_spmm_wide|^k_jku_bwd<20|^k_jku_fwd<20|^k_gemm_f32<2, 2, 1, 1|^k_gemm_f32<4, 1, 1, 1|^k_gemm_f32_shortk<2, 2, 2, 2|^k_sage_wide|^k_sage_gram|^k_sage_rinv|^k_bn_act|^k_bn_bwd_reduce<4, 5|^k_softmax|^k_adj_prep|^k_gemm_fixup|^k_gemm_split|^k_gemm_half|^k_gemm_absmax'
grep -E "$K" gpurun_out/r06_sq_pmc.txt > gpurun_out/r06_bench_c3_pmc_sq.txt
grep -E "$K" gpurun_out/r06_sqs_pmc.txt > gpurun_out/r06_bench_c3_split_pmc_sq.txt
grep -E "$K" gpurun_out/r06_sqh_pmc.txt > gpurun_out/r06_bench_c3_half_pmc_sq.txt
python profiles/make_counters_json.py gpurun_out/r06_bench_c3_pmc_sq.txt gpurun_out/r06_bench_c3_split_pmc_sq.txt gpurun_out/r06_bench_c3_half_pmc_sq.txt > gpurun_out/r06_counters.json
bash tools/pmc_tcc.sh r06_tcc -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-split-leg > /dev/null 2>&1
bash tools/pmc_tcc.sh r06_tccs -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-split-leg --gemm-mode 1 > /dev/null 2>&1
bash tools/pmc_tcc.sh r06_tcch -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-split-leg --gemm-mode 2 > /dev/null 2>&1
python profiles/make_traffic_json.py gpurun_out/r06_tcc_FETCH_SIZE gpurun_out/r06_tcc_WRITE_SIZE gpurun_out/r06_tccs_FETCH_SIZE gpurun_out/r06_tccs_WRITE_SIZE gpurun_out/r06_tcch_FETCH_SIZE gpurun_out/r06_tcch_WRITE_SIZE > gpurun_out/r06_traffic.json
python profiles/summarize_pmc.py gpurun_out/r06_tcc_FETCH_SIZE gpurun_out/r06_tcc_WRITE_SIZE gpurun_out/r06_tcc_TCC --match "k_gemm_f32<2, 2, 2, 2" > gpurun_out/r06_bench_c3_pmc_traffic.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/r06_tcc_FETCH_SIZE gpurun_out/r06_tcc_WRITE_SIZE gpurun_out/r06_tcc_TCC --match "k_spmm_wide" >> gpurun_out/r06_bench_c3_pmc_traffic.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/r06_tccs_FETCH_SIZE gpurun_out/r06_tccs_WRITE_SIZE gpurun_out/r06_tccs_TCC --match "k_gemm_split" >> gpurun_out/r06_bench_c3_pmc_traffic.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/r06_tcch_FETCH_SIZE gpurun_out/r06_tcch_WRITE_SIZE gpurun_out/r06_tcch_TCC --match "k_gemm_half" >> gpurun_out/r06_bench_c3_pmc_traffic.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/r06_tcch_FETCH_SIZE gpurun_out/r06_tcch_WRITE_SIZE gpurun_out/r06_tcch_TCC --match "k_gemm_absmax" >> gpurun_out/r06_bench_c3_pmc_traffic.txt 2>&1
rm -rf gpurun_out/r06_tcc_FETCH_SIZE gpurun_out/r06_tcc_WRITE_SIZE gpurun_out/r06_tcc_TCC gpurun_out/r06_tccs_FETCH_SIZE gpurun_out/r06_tccs_WRITE_SIZE gpurun_out/r06_tccs_TCC gpurun_out/r06_tcch_FETCH_SIZE gpurun_out/r06_tcch_WRITE_SIZE gpurun_out/r06_tcch_TCC
NSTEPS=8 bash tools/prof_step.sh r06_c5 --nodes 8000 --feat 64 --maxn 16000 --steps 6 --warmup 2 --pool 2 --no-split-leg > /dev/null 2>&1
CGC_GEMM_SPLIT_BF16=0 CGC_NATIVE=0 python tools/gemm_time_shapes.py 32 > gpurun_out/r06_gemm_calls_by_shape.txt 2>&1
python tools/split_gemm_bench.py 20 > gpurun_out/r06_split_gemm_standalone.txt 2>&1
python tools/split_gemm_bench.py 20 4 >> gpurun_out/r06_split_gemm_standalone.txt 2>&1
python tools/split_gemm_ksweep.py 20 > gpurun_out/r06_split_gemm_ksweep.txt 2>&1
python tools/eval_bench.py 32 160 > gpurun_out/r06_eval.txt 2>&1
python tools/eval_bench.py 4 64 >> gpurun_out/r06_eval.txt 2>&1
python tools/golden_fp64_report.py > gpurun_out/r06_fp64_report.txt 2>&1
python tools/golden_fp64_report.py --big-route > gpurun_out/r06_fp64_bigroute_exact.txt 2>&1
python tools/golden_fp64_report.py --big-route --split > gpurun_out/r06_fp64_bigroute_split.txt 2>&1
python tools/golden_fp64_report.py --big-route --half > gpurun_out/r06_fp64_bigroute_half.txt 2>&1
python tools/decision_noise.py medium_plain tiny_plain medium_shipped > gpurun_out/r06_decision_noise.txt 2>&1
python tools/jk_bench.py > gpurun_out/r06_jk_bench.txt 2>&1
python tools/operand_range.py 32 > gpurun_out/r06_operand_range.txt 2>&1
(for b in 32 4; do for m in 1 0; do echo "== batch $b, CGC_GRAPH_LOCAL=$m"; CGC_GRAPH_LOCAL=$m python bench.py --batch $b --no-cpu-baseline --no-split-leg --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'graphs/s', d['ms_per_step'], 'ms/step')"; done; done) > gpurun_out/r06_graph_local_ab.txt 2>&1
python bench.py --no-cpu-baseline --spatial > gpurun_out/r06_spatial.json 2>/dev/null
cp gpurun_out/r06_counters.json gpurun_out/r06_traffic.json profiles/      # (on the box: the line below then quotes this run's own counter summaries)
python bench.py > gpurun_out/r06_default_last.json 2> gpurun_out/r06_default_last.err      # (after the counter passes: nothing else is meant to differ)
# timing-only ablations (variant libraries built beforehand by tools/ablate_presplit.sh / tools/variant_lib.sh; skipped when absent)
V=$R/cgc-net_amd/csrc/variants
if [ -f $V/libcgc_abl_valu.so ]; then
  (for v in "" abl_valu abl_valu_lds ""; do if [ -z "$v" ]; then unset CGC_LIB; echo "== k_gemm_split as built"; else export CGC_LIB=$V/libcgc_$v.so; echo "== $v"; fi; SPLIT_BENCH_MODES=1 python tools/split_gemm_bench.py 20 2>&1 | grep -v amdgpu.ids | tail -7 | cut -c1-150; done; unset CGC_LIB) > gpurun_out/r06_split_gemm_presplit_ablation.txt 2>&1
fi
if [ -f $V/libcgc_h_none.so ]; then
  (for v in "" h_nosplit h_nowrite h_nofrag h_noload h_nobar h_none ""; do if [ -z "$v" ]; then unset CGC_LIB; echo "== k_gemm_half as built"; else export CGC_LIB=$V/libcgc_$v.so; echo "== $v"; fi; SPLIT_BENCH_MODES=2 SPLIT_BENCH_CASES=1,4 python tools/split_gemm_bench.py 20 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-150; done; unset CGC_LIB) > gpurun_out/r06_half_gemm_ablation.txt 2>&1
fi
ls gpurun_out | grep r06_ | head -100
cat gpurun_out/r06_configurations_raw.txt | tail -12
