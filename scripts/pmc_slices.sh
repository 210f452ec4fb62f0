#!/bin/bash
# SQ counters for the stencil-slice SpMV (separate passes, kernel-trace only).  usage: pmc_slices.sh <tag> [xcd/dbg] [blocks/CU]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcs_$1
export PG_SPMV_XCD=${2:-1} PG_SPMV_BLOCKS_PER_CU=${3:-5}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d ${out}_$i -- python3 scripts/spmv_time_only.py 512 > ${out}_$i.log 2>&1
  echo "pass $i done" 
done
for i in 1 2 3; do python3 scripts/pmc_parse.py "${out}_$i/*/*counter_collection.csv" "k_spmv_s"; done > gpurun_out/pmcs_$1.txt 2>&1
cat gpurun_out/pmcs_$1.txt
