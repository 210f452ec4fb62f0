#!/bin/bash
# A/B of two library builds on the bench workload: scripts/ab_bench.sh <libA> <libB> [steps]
steps=${3:-60}
for rep in 1 2; do
  for lib in $1 $2; do
    PG_LIB_PATH=$PWD/$lib python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), d['config']['krylov_iters_per_step'], round(d['roofline']['avg_launch_ms'],4))"
  done
done
