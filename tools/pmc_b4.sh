#!/bin/bash
# On the GPU box: instruction / cycle / LDS counters of the serial loop's kernels (tools/b4_serial_loop.py, 4 views), three --pmc passes.
#   tools/pmc_b4.sh <tag>   -> gpurun_out/<tag>_pmcb4_<group>.txt
tag=$1
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=$PWD/gpurun_out/pmcb4_${tag}_$i
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $grp -d "$out" -o pmc --output-format csv -- python3 tools/b4_serial_loop.py 4 > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_fold.py "$f" | grep "k_backward_sq\|k_render" > gpurun_out/${tag}_pmcb4_$i.txt && cat gpurun_out/${tag}_pmcb4_$i.txt
done
