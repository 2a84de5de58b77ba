#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "wide_spmm" > gpurun_out/r05_k4_tests.log 2>&1; tail -3 gpurun_out/r05_k4_tests.log
python tools/spmm_patch_bench.py 32 1800 > gpurun_out/r05_k4_bench.txt 2>&1
python tools/spmm_patch_bench.py 4 1800 >> gpurun_out/r05_k4_bench.txt 2>&1
python tools/spmm_patch_bench.py 8 8000 1600 >> gpurun_out/r05_k4_bench.txt 2>&1
cat gpurun_out/r05_k4_bench.txt
rm -f gpurun_out/r05_split_gemm_error_table.txt
timeout 600 python -m pytest tests/test_evalio_gpu.py tests/test_split_gemm_gpu.py -q > gpurun_out/r05_k4_tests2.log 2>&1; tail -3 gpurun_out/r05_k4_tests2.log
