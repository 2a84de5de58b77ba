#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/split_gemm_ksweep.py 20 > gpurun_out/r05_split_ksweep.txt 2>&1; cat gpurun_out/r05_split_ksweep.txt
for v in $(ls cgc-net_amd/csrc/variants/ | grep xs_); do
  echo "== $v"; CGC_LIB=$PWD/cgc-net_amd/csrc/variants/$v SPLIT_BENCH_CASES=1,5 timeout 120 python tools/split_gemm_bench.py 20 2>&1 | grep -E "split" | cut -c1-60,105-200
done > gpurun_out/r05_split_ablation.txt 2>&1
echo "== default"; SPLIT_BENCH_CASES=1,5 timeout 120 python tools/split_gemm_bench.py 20 2>&1 | grep -E "split" | cut -c1-60,105-200 >> gpurun_out/r05_split_ablation.txt
cat gpurun_out/r05_split_ablation.txt
