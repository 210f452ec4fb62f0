#!/bin/bash
# strip width of the SpMV's work items against the grid size (PG_SPMV_STRIP = grid lines per strip)
for n in 768 512; do
for st in $( [ $n = 768 ] && echo "4 6 8 10 12 16" || echo "8 12 16 20 24" ); do
  PG_SPMV_STRIP=$st python bench.py --n $n --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$n strip=$st', round(d['value'],1), 'launch_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'frac', round(d['roofline']['frac_events'],3))"
done; done
