#!/bin/bash
# Memory-side counters of the slice SpMV in the solver loop (separate passes, kernel-trace only; few counters per pass:
# a request the hardware cannot schedule aborts rocprofv3).  usage: pmc_memside.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcm_$1
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d ${out}_$i -- python3 scripts/dev_perf.py 512 2 > ${out}_$i.log 2>&1
  echo "pass $i rc=$?" | tee -a gpurun_out/pmcm_$1_progress.txt
done
for i in 1 2 3 4; do python3 scripts/pmc_parse.py "${out}_$i/*/*counter_collection.csv" "k_spmv_s"; done > gpurun_out/pmcm_$1.txt 2>&1
for i in 1 2 3 4; do python3 scripts/pmc_parse.py "${out}_$i/*/*counter_collection.csv" "k_bicg_xrp"; done > gpurun_out/pmcm_$1_xrp.txt 2>&1
cat gpurun_out/pmcm_$1.txt; echo ---- xrp; cat gpurun_out/pmcm_$1_xrp.txt
