#!/bin/bash
# 768^3 (past the Infinity Cache): does the work split / occupancy of the SpMV matter there?
for cfg in "PG_SPMV_BLOCKS_PER_CU=0" "PG_SPMV_BLOCKS_PER_CU=3" "PG_SPMV_BLOCKS_PER_CU=6" "PG_SPMV_BLOCKS_PER_CU=8" "PG_SPMV_STRIP=8" "PG_SPMV_STRIP=32" "PG_KRYLOV_NT=0"; do
  env $cfg python bench.py --n 768 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), 'launch_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'frac', round(d['roofline']['frac_events'],3))"
done
