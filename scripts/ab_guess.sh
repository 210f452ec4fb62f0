#!/bin/bash
# A/B of the extrapolated start on the bench workload: GUESS_SETS = list of "states:depth" (older states read at most :
# older states kept); "0:1" = off
for steps in ${GUESS_STEPS:-20 200}; do
  for rep in ${GUESS_REPS:-1 2}; do
    for k in ${GUESS_SETS:-0:1 4:7}; do
      export PG_GUESS_STATES=${k%%:*} PG_GUESS_DEPTH=${k##*:}
      python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('states:depth=$k steps=$steps', round(d['value'],2), 'deg', d['config']['polynomial_preconditioner_degree'], 'its', d['config']['krylov_iters_per_step'], 'launch_ms', round(d['roofline']['avg_launch_ms'],4))"
    done
  done
done
