#!/bin/bash
# On the GPU box: instruction / cycle / LDS counters of the library's kernels under any target (three --pmc passes).
#   tools/pmc_any.sh <tag> <grep pattern of kernel names> <python script> [args]   -> gpurun_out/<tag>_pmc_<group>.txt
tag=$1; pat=$2; shift 2
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=$PWD/gpurun_out/pmcany_${tag}_$i
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $grp -d "$out" -o pmc --output-format csv -- python3 "$@" > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_fold.py "$f" | grep "$pat" > gpurun_out/${tag}_pmc_$i.txt && cat gpurun_out/${tag}_pmc_$i.txt
done
