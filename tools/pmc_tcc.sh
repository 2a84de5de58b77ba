#!/bin/bash
# HBM-traffic counter passes (one counter group per run, each bounded by `timeout`): FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS
# usage: tools/pmc_tcc.sh <tag> -- <command...>   -> gpurun_out/<tag>_{FETCH_SIZE,WRITE_SIZE,TCC}/p_counter_collection.csv
tag=$1; shift 2
export TMPDIR=/tmp
R=$(pwd)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1 | sed 's/_sum//; s/TCC_HIT/TCC/')
  d=$R/gpurun_out/${tag}_$name
  rm -rf $d
  ( cd /tmp && timeout 420 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d/raw -o p -- "$@" > $R/gpurun_out/${tag}_$name.log 2>&1 )
  f=$(find $d/raw -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $d/p_counter_collection.csv; fi
  rm -rf $d/raw
  ls -la $d
done
