#!/bin/bash
# Round 6: the GPU test suite (error table of the split GEMM regenerated) + the default bench line without the CPU leg
mkdir -p gpurun_out
rm -f gpurun_out/r06_split_gemm_error_table.txt gpurun_out/r06_half_gemm_error_table.txt
rm -f gpurun_out/r06_discrete_decisions.txt; CGC_DECISION_LOG=$PWD/gpurun_out/r06_discrete_decisions.txt CGC_SPLIT_ERROR_TABLE=$PWD/gpurun_out/r06_split_gemm_error_table.txt CGC_HALF_ERROR_TABLE=$PWD/gpurun_out/r06_half_gemm_error_table.txt timeout 3000 python -m pytest tests -m gpu -q --durations=14 > gpurun_out/r06_gputests.log 2>&1; echo rc=$? >> gpurun_out/r06_gputests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|s call|s setup|Error|assert" gpurun_out/r06_gputests.log | cut -c1-300 | tail -40
