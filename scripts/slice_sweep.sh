#!/bin/bash
# launch / slicing parameters of the stencil-slice SpMV on the 512^3 system (time only)
for cfg in "1 4 16" "1 5 16" "1 4 24" "257 4 16" "513 4 16"; do
  set -- $cfg
  echo "== xcd/dbg $1 blocks/CU $2 minrun $3"
  PG_SPMV_XCD=$1 PG_SPMV_BLOCKS_PER_CU=$2 PG_SPMV_MINRUN=$3 PG_DEBUG=1 python scripts/spmv_time_only.py 512 2>&1 | grep -E "spmv" | tail -2
done
