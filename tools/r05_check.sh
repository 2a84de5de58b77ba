#!/bin/bash
# Round 5, one GPU call: the whole -m gpu suite (split-mode tests and the parametrised full-size parity tests included), the default bench
# line (exact headline + split leg), the six dominant products stand-alone in both modes, the fused adjacency backward against the
# reference's float64 fixtures.   -> gpurun_out/r05_*
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r05_split_gemm_error_table.txt
CGC_SPLIT_ERROR_TABLE=$PWD/gpurun_out/r05_split_gemm_error_table.txt timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests.log 2>&1; echo rc=$? >> gpurun_out/r05_gputests.log
tail -4 gpurun_out/r05_gputests.log
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; cat gpurun_out/r05_bench_default.json
python bench.py --batch 4 --no-cpu-baseline > gpurun_out/r05_bench_b4.json 2>/dev/null; cat gpurun_out/r05_bench_b4.json
timeout 300 python tools/split_gemm_bench.py 20 > gpurun_out/r05_split_bench.txt 2>&1; cat gpurun_out/r05_split_bench.txt
CGC_ADJ_FUSED=1 python tools/golden_fp64_report.py 2>&1 | grep -v "^  " > gpurun_out/r05_fp64_report_adj_fused.txt; cat gpurun_out/r05_fp64_report_adj_fused.txt
CGC_ADJ_FUSED=1 python bench.py --no-cpu-baseline --no-split-leg > gpurun_out/r05_bench_adj_fused.json 2>/dev/null; cat gpurun_out/r05_bench_adj_fused.json
