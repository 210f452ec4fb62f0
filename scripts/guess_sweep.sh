#!/bin/bash
# the degree estimator with the extrapolated start: value and the number of solves whose degree fell short
for steps in ${SWEEP_STEPS:-20 120 600}; do
for cfg in "PG_POLY_TREND=1" "PG_POLY_TREND=0"; do
  env $cfg python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg steps=$steps', round(d['value'],2), 'short', d['config']['solves_that_needed_a_second_application'], 'deg_mean', round(d['step_roofline']['polynomial_preconditioner_degree_mean'],2), 'products/step', d['config']['products_per_step_in_the_solves'])"
done
done
