#!/bin/bash
# A/B of environment settings on the bench workload: scripts/ab_env.sh "<ENV=..> ..." "<ENV=..> ..." [steps ...]
a=$1; b=$2; shift 2
for steps in ${@:-20 200}; do
  for rep in 1 2; do
    for cfg in "$a" "$b"; do
      env $cfg python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg steps=$steps', round(d['value'],2), 'its', round(d['config']['krylov_iters_per_step'],3), 'deg_mean', round(d['step_roofline']['polynomial_preconditioner_degree_mean'],2), 'launch_us', round(d['roofline']['avg_launch_ms']*1e3,2))"
    done
  done
done
