#!/bin/bash
# copy the round-6 measurements (tools/r06_measure.sh -> gpurun_out/r06_*) into profiles/ under the names DESIGN.md cites
G=gpurun_out; P=profiles
cp $G/r06_default.json $P/r06_bench_default.json
cp $G/r06_b4.json $P/r06_bench_b4_line.json
cp $G/r06_c5_draw.json $P/r06_c5_line.json
cp $G/r06_c3_trace.txt $P/r06_bench_c3_shipped_kernel_trace.txt
cp $G/r06_c3_split_trace.txt $P/r06_bench_c3_split_kernel_trace.txt
cp $G/r06_b4_trace.txt $P/r06_bench_c3_batch4_kernel_trace.txt
cp $G/r06_c5_trace.txt $P/r06_c5_draw_order_kernel_trace.txt
(echo "# GPU idle time inside steady-state steps (rocprofv3 kernel trace of bench.py --steps 10 --warmup 3; profiles/gaps_rocpd.py), round 6"; echo "## 32 graphs, default"; cat $G/r06_c3_gaps.txt; echo "## 32 graphs, Python's cyclic garbage collector off during the timed steps (bench.py --no-gc)"; cat $G/r06_c3_nogc_gaps.txt; echo "## 4 graphs per GPU"; cat $G/r06_b4_gaps.txt) > $P/r06_step_gaps.txt
cp $G/r06_bench_c3_pmc_sq.txt $G/r06_bench_c3_split_pmc_sq.txt $G/r06_bench_c3_pmc_traffic.txt $G/r06_counters.json $G/r06_traffic.json $P/
cp $G/r06_gemm_calls_by_shape.txt $G/r06_split_gemm_standalone.txt $G/r06_split_gemm_ksweep.txt $P/
cp $G/r06_eval.txt $P/r06_eval_throughput.txt
(echo "# HIP path vs the REFERENCE's float64 gradients (tests/golden/*_fp64.npz), all eight fixtures, DEFAULT routing (no product of these fixtures reaches the split kernel), bar max(1e-4, ulp64) -- 1e-4 on every parameter here; tools/golden_fp64_report.py on 1xMI355X, round 6"; grep -v "amdgpu.ids" $G/r06_fp64_report.txt | grep -v "^  ") > $P/r06_gradients_vs_reference_fp64.txt
(echo "# round 6 configurations, 1xMI355X (tools/final_measure.sh r06): headline = exact fp32 GEMM; 'split mode' = the second leg of the same bench run with the six dominant products in CGC_GEMM_SPLIT_BF16"; tail -10 $G/r06_configurations_raw.txt) > $P/r06_configurations.txt
if [ -f $G/r06_split_gemm_error_table.txt ]; then (echo "# cgc_gemm_f32_ws mode CGC_GEMM_SPLIT_BF16 (csrc/gemm_split.hip) next to the exact fp32 MFMA kernel, every form the step uses; error of every output against float64 relative to sum_k |a||b|; inputs: normal = N(0,1); wide = every OUTPUT row / column scaled by 2^-30..2^+30; skewk = the same scales along K in both operands (one or two terms are the sum); tiny = scaled by 2^-100.  tests/test_split_gemm_gpu.py on 1xMI355X, round 6"; cat $G/r06_split_gemm_error_table.txt) > $P/r06_split_gemm_error_table.txt; fi
(echo "# the same with EVERY product forced onto the 128 x 128 route (cgc_gemm_tuning(11)), exact fp32 GEMM, then split bf16 (k_gemm_split): tools/golden_fp64_report.py --big-route [--split], round 6.  Bar: max(1e-4, ulp64) -- the two parameters above 1e-4 are the two whose own conditioning (ulp64 = 1.7e-4 / 2.5e-4) exceeds it"; grep -v "amdgpu.ids" $G/r06_fp64_bigroute_exact.txt | grep -v "^  "; grep -v "amdgpu.ids" $G/r06_fp64_bigroute_split.txt | grep -v "^  ") > $P/r06_gradients_vs_reference_fp64_big_route.txt
(echo "# discrete decisions the HIP path took differently from float64, per comparison of the GPU suite (tests/discrete.py, CGC_DECISION_LOG), round 6: what MAX_TIE / RELU_TIE rest on"; sort -u $G/r06_discrete_decisions.txt) > $P/r06_discrete_decisions.txt
(echo "# tools/jk_bench.py (DenseJK kernels stand-alone at the three level sizes of C3), round 6: v_rcp_f32 + one Newton step in the gates"; grep -v amdgpu $G/r06_jk_bench.txt) > $P/r06_jk_bench.txt
(echo "# bench.py --no-split-leg with the graph structure built graph by graph (cgc_graph_build_local, 2 launches) and by the general build (CGC_GRAPH_LOCAL=0, ~20 launches), same box, round 6"; cat $G/r06_graph_local_ab.txt; grep -E "graph_local|k_hist_rows|k_fill_rows|k_sort|k_hist_cols|k_fill_cols|k_scan|k_compact|k_zero_ints|k_edge_renorm|k_transpose_vals|k_csr_invdeg" $G/r06_c3_trace.txt | cut -c1-200) > $P/r06_graph_local_build.txt
(echo "# tools/operand_range.py 32 on 1xMI355X, round 6: where the operands of the step's six dominant products sit between the input families of tests/test_split_gemm_gpu.py"; grep -v amdgpu $G/r06_operand_range.txt) > $P/r06_operand_range_along_k.txt
