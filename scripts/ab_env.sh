#!/bin/bash
# A/B of environment settings on the bench workload: scripts/ab_env.sh "<ENV=..> ..." "<ENV=..> ..." [steps]
steps=${3:-40}
for rep in 1 2; do
  for cfg in "$1" "$2"; do
    env $cfg python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],2), d['config']['krylov_iters_per_step'], round(d['roofline']['avg_launch_ms'],4))"
  done
done
