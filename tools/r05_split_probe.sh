#!/bin/bash
# round 5: the split-bf16 GEMM -- parity tests, timing ablations (variant libraries), counters.  Runs on the GPU box.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_split_gemm_gpu.py tests/test_bench_size_parity_gpu.py -x -q -s > gpurun_out/r05_split_tests.log 2>&1; echo rc=$? >> gpurun_out/r05_split_tests.log
grep -E "exact max|passed|failed|rc=" gpurun_out/r05_split_tests.log | tail -80
timeout 200 python tools/split_gemm_bench.py 20 > gpurun_out/r05_split_bench.txt 2>&1; cat gpurun_out/r05_split_bench.txt
for v in $(ls cgc-net_amd/csrc/variants/ | grep xs_); do
  echo "== $v"; CGC_LIB=$PWD/cgc-net_amd/csrc/variants/$v SPLIT_BENCH_CASES=${ABL_CASES:-1,5} timeout 120 python tools/split_gemm_bench.py 20 2>&1 | grep -E "split|lib="
done > gpurun_out/r05_split_ablation.txt 2>&1
cat gpurun_out/r05_split_ablation.txt
PMC_PASSES=3 SPLIT_BENCH_CASES=1,5 tools/pmc_run.sh r05_split "k_gemm_split" -- python $PWD/tools/split_gemm_bench.py 3 > /dev/null 2>&1
cat gpurun_out/r05_split_pmc.txt
