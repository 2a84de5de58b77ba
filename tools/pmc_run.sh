#!/bin/bash
# rocprofv3 counter passes (SQ: 8 slots per pass) around one command; per-kernel medians -> gpurun_out/<tag>_pmc.txt
# usage: tools/pmc_run.sh <tag> <match> -- <command...>
tag=$1; match=$2; shift 3
export TMPDIR=/tmp
R=$(pwd)
rm -f gpurun_out/${tag}_pmc.txt
i=0
passes=${PMC_PASSES:-5}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "WRITE_SIZE"; do
  i=$((i+1))
  if [ $i -gt $passes ]; then break; fi
  d=$R/gpurun_out/${tag}_pmc$i
  ( cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- "$@" > $R/gpurun_out/${tag}_pmc$i.log 2>&1 )
  ls -R $d >> gpurun_out/${tag}_pmc$i.log 2>&1; f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    mkdir -p $d/flat; cp $f $d/flat/p_counter_collection.csv
    python profiles/summarize_pmc.py $d/flat --match "$match" >> gpurun_out/${tag}_pmc.txt 2>&1
  else
    echo "pass $i failed: $(tail -3 gpurun_out/${tag}_pmc$i.log | tr '\n' ' ')" >> gpurun_out/${tag}_pmc.txt
  fi
  rm -rf $d
done
cat gpurun_out/${tag}_pmc.txt
