#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r05_split_gemm_error_table.txt
CGC_SPLIT_ERROR_TABLE=$PWD/gpurun_out/r05_split_gemm_error_table.txt timeout 1700 python -m pytest tests -m gpu -q --durations=14 > gpurun_out/r05_gputests.log 2>&1; echo rc=$? >> gpurun_out/r05_gputests.log
grep -E "passed|failed|^FAILED|rc=|s call|s setup" gpurun_out/r05_gputests.log | cut -c1-200 | tail -24
