timeout -k 10 500 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests_poly.log 2>&1; echo tests_rc=$?; tail -4 gpurun_out/gpu_tests_poly.log
run() { python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/ab_poly_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms']*1e3,1), d['config']['krylov_iters_per_step'], round(d['step_roofline']['frac'],3), d['step_roofline']['neumann_preconditioner'], round(d['step_roofline']['gershgorin_radius'],3))"; }
for rep in 1 2; do
PG_POLY=0 run plain
run neumann
done
