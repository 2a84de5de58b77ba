#!/bin/bash
# Round 5, last pass on the final kernels: the whole -m gpu suite, then tools/r05_measure.sh, then this round's stand-alone probes
# (wide SAGE forward in both routes, what a 1-read + 1-write pass reaches)  -> gpurun_out/r05_*
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r05_split_gemm_error_table.txt
CGC_SPLIT_ERROR_TABLE=$PWD/gpurun_out/r05_split_gemm_error_table.txt timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests.log 2>&1; echo rc=$? >> gpurun_out/r05_gputests.log
tail -3 gpurun_out/r05_gputests.log
bash tools/r05_measure.sh > gpurun_out/r05_measure.log 2>&1
(python tools/sage_wide_bench.py; CGC_SAGE_WIDE_COLS=0 python tools/sage_wide_bench.py; python tools/sage_wide_bench.py 24000 20 1600; CGC_SAGE_WIDE_COLS=0 python tools/sage_wide_bench.py 24000 20 1600) 2>&1 | grep "^n " > gpurun_out/r05_sage_wide.txt
python tools/stream_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_stream.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_rows_probe tools/probes/stream_rows_probe.hip && /tmp/stream_rows_probe >> gpurun_out/r05_stream.txt 2>&1
tail -12 gpurun_out/r05_measure.log
(python tools/thin_probe.py; THIN_COLD=1 python tools/thin_probe.py) 2>&1 | grep "^N \|^read" > gpurun_out/r05_thin_products.txt
