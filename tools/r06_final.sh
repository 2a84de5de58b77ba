#!/bin/bash
# Round 6, last pass on the final kernels: the whole -m gpu suite, then tools/r06_measure.sh  -> gpurun_out/r06_*
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/r06_suite.sh
bash tools/r06_measure.sh > gpurun_out/r06_measure.log 2>&1
tail -14 gpurun_out/r06_measure.log
