#!/bin/bash
# usage: scripts/spmv_sweep.sh "<variant:blocks_per_cu> ..."  [n]
n=${2:-512}
for vb in $1; do
  v=${vb%%:*}; b=${vb##*:}
  echo "== variant $v blocks/CU $b"
  PG_SPMV_VARIANT=$v PG_SPMV_BLOCKS_PER_CU=$b python scripts/dev_perf.py $n 5 2>&1 | tail -2
done
