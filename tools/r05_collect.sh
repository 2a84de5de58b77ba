#!/bin/bash
# copy the round-5 measurements (tools/r05_measure.sh -> gpurun_out/r05_*) into profiles/ under the names DESIGN.md cites
G=gpurun_out; P=profiles
cp $G/r05_default.json $P/r05_bench_default.json
cp $G/r05_b4.json $P/r05_bench_b4_line.json
cp $G/r05_c5_draw.json $P/r05_c5_line.json
cp $G/r05_c3_trace.txt $P/r05_bench_c3_shipped_kernel_trace.txt
cp $G/r05_c3_split_trace.txt $P/r05_bench_c3_split_kernel_trace.txt
cp $G/r05_b4_trace.txt $P/r05_bench_c3_batch4_kernel_trace.txt
cp $G/r05_c5_trace.txt $P/r05_c5_draw_order_kernel_trace.txt
(echo "# GPU idle time inside steady-state steps (rocprofv3 kernel trace of bench.py --steps 10 --warmup 3; profiles/gaps_rocpd.py), round 5"; echo "## 32 graphs, default"; cat $G/r05_c3_gaps.txt; echo "## 32 graphs, Python's cyclic garbage collector off during the timed steps (bench.py --no-gc)"; cat $G/r05_c3_nogc_gaps.txt; echo "## 4 graphs per GPU"; cat $G/r05_b4_gaps.txt) > $P/r05_step_gaps.txt
cp $G/r05_bench_c3_pmc_sq.txt $G/r05_bench_c3_split_pmc_sq.txt $G/r05_bench_c3_pmc_traffic.txt $G/r05_counters.json $G/r05_traffic.json $P/
cp $G/r05_gemm_calls_by_shape.txt $G/r05_split_gemm_standalone.txt $G/r05_split_gemm_ksweep.txt $P/
cp $G/r05_eval.txt $P/r05_eval_throughput.txt
cp $G/r05_adj_fused.json $P/r05_bench_adj_fused_line.json; cp $G/r05_spatial.json $P/r05_bench_spatial_line.json
(echo "# HIP path vs the REFERENCE's float64 gradients (tests/golden/*_fp64.npz), all eight fixtures, bar 1e-4 strict (no ulp64 widening); tools/golden_fp64_report.py on 1xMI355X, round 5"; grep -v "amdgpu.ids" $G/r05_fp64_report.txt | grep -v "^  ") > $P/r05_gradients_vs_reference_fp64.txt
(echo "# the same with the opt-in fused adjacency backward (CGC_ADJ_FUSED=1; row term in double, two k columns): tiny_elu 1.1e-4, medium_plain 1.0e-4 -- at / over the bar: stays opt-in"; grep -v "amdgpu.ids" $G/r05_fp64_report_adj_fused.txt | grep -v "^  ") > $P/r05_gradients_vs_reference_fp64_adj_fused.txt
(echo "# round 5 configurations, 1xMI355X (tools/final_measure.sh r05): headline = exact fp32 GEMM; 'split mode' = the second leg of the same bench run with the six dominant products in CGC_GEMM_SPLIT_BF16"; tail -10 $G/r05_configurations_raw.txt) > $P/r05_configurations.txt
if [ -f $G/r05_split_gemm_error_table.txt ]; then (echo "# cgc_gemm_f32_ws mode CGC_GEMM_SPLIT_BF16 (csrc/gemm_split.hip) next to the exact fp32 MFMA kernel, every form the step uses; error of every output against float64 relative to sum_k |a||b|; inputs: normal = N(0,1); wide = every OUTPUT row / column scaled by 2^-30..2^+30; skewk = the same scales along K in both operands (one or two terms are the sum); tiny = scaled by 2^-100.  tests/test_split_gemm_gpu.py on 1xMI355X, round 5"; cat $G/r05_split_gemm_error_table.txt) > $P/r05_split_gemm_error_table.txt; fi
