#!/bin/bash
# usage: scripts/pmc_fetch.sh <variant> <blocks_per_cu> <tag>   -> FETCH_SIZE / WRITE_SIZE of the SpMV kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PG_SPMV_VARIANT=$1 PG_SPMV_BLOCKS_PER_CU=$2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_$3 -- python3 scripts/dev_perf.py 512 2 > gpurun_out/pmcf_$3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_$3 -- python3 scripts/dev_perf.py 512 2 > gpurun_out/pmcw_$3.log 2>&1
